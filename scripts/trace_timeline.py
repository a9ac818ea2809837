"""Busy fraction of the GPU per 0.5-ms bin over the last training step of a rocprofv3 --kernel-trace csv of
bench.py, with the kernel that dominates each bin: where the step is host-bound (short kernels with gaps).
usage: trace_timeline.py <dir with *_kernel_trace.csv> <warmup> <steps> [bin_us]"""
import collections, csv, glob, re, sys
d, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bin_ns = int(float(sys.argv[4]) * 1000) if len(sys.argv) > 4 else 500000
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)   # (child processes leave small traces of their own)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
gi = [i for i, r in enumerate(rows) if "gemm_f32_" in r[2]]
per = len(gi) // (warmup + steps)
lo = gi[(warmup + steps - 2) * per]          # first GEMM of the second-to-last step ...
hi = gi[(warmup + steps - 1) * per]          # ... to the first GEMM of the last: one whole step
region = rows[lo:hi]
t0 = region[0][0]
short = lambda k: re.sub(r"^void |\(anonymous namespace\)::|at::native::|<.*|\(.*", "", k)[:48]
busy = collections.defaultdict(float)
who = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for s, e, k in region:
    cnt[(s - t0) // bin_ns] += 1
    t = s
    while t < e:
        b = (t - t0) // bin_ns
        nxt = min(e, t0 + (b + 1) * bin_ns)
        busy[b] += nxt - t
        who[b][short(k)] += nxt - t
        t = nxt
print("one step = %.2f ms; per %.1f-ms bin: busy %%, kernels started, dominant kernel" % ((region[-1][1] - t0) / 1e6, bin_ns / 1e6))
for b in range(0, max(busy) + 1):
    dom = max(who[b].items(), key=lambda kv: kv[1])[0] if who[b] else "-"
    print("%7.1f ms  %5.1f %%  %4d  %s" % (b * bin_ns / 1e6, 100.0 * busy[b] / bin_ns, cnt[b], dom))
