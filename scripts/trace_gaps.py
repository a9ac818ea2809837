"""Where is the GPU idle inside a training step?  From a rocprofv3 --kernel-trace csv of bench.py:
the largest gaps between consecutive kernels of the timed region, with the kernels either side.
usage: trace_gaps.py <dir with *_kernel_trace.csv> <warmup> <steps> [min_gap_us]"""
import collections, csv, glob, sys
d, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
min_gap = float(sys.argv[4]) if len(sys.argv) > 4 else 15.0
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)   # (child processes leave small traces of their own)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
gi = [i for i, r in enumerate(rows) if "gemm_f32_" in r[2]]
per = len(gi) // (warmup + steps)
region = rows[gi[warmup * per]:]
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:70]
gaps = collections.defaultdict(lambda: [0.0, 0])
end = region[0][1]
tot = 0.0
for (s, e, k), (ps, pe, pk) in zip(region[1:], region[:-1]):
    g = (s - end) / 1e3
    if g > 0:
        tot += g
        if g >= min_gap:
            a = gaps[(short(pk), short(k))]
            a[0] += g; a[1] += 1
    end = max(end, e)
print("idle total %.2f ms/step; gaps >= %.0f us:" % (tot / steps / 1e3, min_gap))
for (a, b), (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.1f us/step  x%-3d  %s  ->  %s" % (g / steps, n // steps if n >= steps else n, a, b))
