"""Per-kernel register / scratch / LDS usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py ait_amd/csrc/gemm_f32.hip [name-filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-c", "--cuda-device-only", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:],
                   stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
cur = None
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        if cur and flt in cur["name"]:
            print(cur)
        name = subprocess.run(["c++filt", m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = {"name": re.sub(r"ait_gemm::|\(ait_gemm::GemmArgs\)|void ", "", name)}
    else:
        cur[m.group(1).split(" ")[0]] = int(m.group(2))
if cur and flt in cur["name"]:
    print(cur)
