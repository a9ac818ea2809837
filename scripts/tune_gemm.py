"""Build scripts/tune_gemm.hip and time every tile variant on the AIT GEMM shapes (GPU box)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
so = os.path.join(ROOT, "scripts", "_tune_gemm.so")
src = os.path.join(ROOT, "scripts", "tune_gemm.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared",
                           "-fPIC", "-I", os.path.join(ROOT, "include"), src, "-o", so])
if not torch.cuda.is_available():
    print("built", so); sys.exit(0)
L = ctypes.CDLL(so)
vp, i = ctypes.c_void_p, ctypes.c_int
L.tune_gemm.argtypes = [i, i, i, i, i, i, vp, i, vp, i, vp, i, i, i, vp]
names = {0: "128x128x16 4w", 1: "256x128x16 8w", 2: "256x128x16 8w schedbar", 3: "128x128x16 4w schedbar",
         4: "256x128x16 8w 3buf", 5: "256x128x16 8w 3buf midstore", 6: "256x128x16 3buf midstore prio", 7: "256x128x32 3buf midstore",
         8: "256x128x16 3buf row-image", 9: "256x128x16 4 waves of 128x64, 3buf", 10: "128x256x16 4 waves of 64x128, 3buf",
         11: "256x128x16 4 waves of 128x64, 3buf row-image",
         12: "ABLATION v4 without global loads", 13: "ABLATION v4 without global loads / LDS stores",
         14: "ABLATION v4 without loads / stores / barrier", 15: "256x128x16 3buf direct-to-LDS",
         16: "ABLATION v15 with the slab wait relaxed to vmcnt(3)", 17: "256x256x16 16 waves direct-to-LDS",
         18: "128x128x16 4 waves direct-to-LDS, 3 blocks/CU", 19: "v15 with the second resident workgroup delayed half a slab",
         20: "256x128x16 direct-to-LDS, 4 waves of 128x64", 21: "256x256x16 direct-to-LDS, 8 waves of 128x64",
         22: "128x128x32 two-stage direct-to-LDS, 4 waves (the rocBLAS macro tile)", 23: "256x128x32 two-stage direct-to-LDS, 8 waves, 1 block/CU"}
def timeit(fn, n=int(os.environ.get("TUNE_N", "10")), w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 76800
dt = "bf16" if "bf16" in sys.argv else "f32"
shapes = [("qkv NT", M, 1536, 512, 0, 1, 1), ("ffn2 NT", M, 512, 2048, 0, 1, 1), ("ffn1 NT", M, 2048, 512, 0, 1, 1),
          ("dgrad NN", M, 512, 2048, 0, 0, 1), ("dgrad NN", M, 2048, 512, 0, 0, 1),
          ("wgrad TN", 512, 2048, M, 1, 0, 16), ("wgrad TN", 2048, 512, M, 1, 0, 16), ("wgrad TN", 1536, 512, M, 1, 0, 24)]
variants = [int(v) for v in sys.argv[1:]] or list(names)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, m, n, k, ta, tb, sk in shapes:
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    c = torch.zeros(m, n, device="cuda")
    ref = (a.t() if ta else a) @ (b.t() if tb else b)
    line = "%-9s M=%6d N=%5d K=%6d:" % (name, m, n, k)
    for v in variants:
        flags = 4 if sk > 1 else 0
        def f():
            return L.tune_gemm(v, ta, tb, m, n, k, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                               c.data_ptr(), n, flags, sk, st)
        c.zero_(); rc = f(); torch.cuda.synchronize()
        err = float((c - ref).abs().max() / ref.abs().max()) if rc == 0 else -1
        ms = timeit(f) if rc == 0 else float("nan")
        line += "  v%d %6.1f TF%s" % (v, 2.0 * m * n * k / ms / 1e9, "" if err < 1e-4 else "(ERR %.1e rc %d)" % (err, rc))
    print(line, flush=True)
print({k: names[k] for k in variants})
