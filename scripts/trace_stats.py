"""Per-kernel statistics of the TIMED region of a bench.py run from a rocprofv3 --kernel-trace csv.

usage: trace_stats.py <dir with *_kernel_trace.csv> <warmup> <steps> <out.csv>

The warm-up steps (and MIOpen's find phase inside them) are cut off by locating the first GEMM
dispatch of the first timed step: bench.py issues a fixed number of gemm_f32_kernel launches per
step, so the timed region starts at GEMM dispatch number warmup * (n_gemm / (warmup + steps))."""
import collections, csv, glob, sys
d, warmup, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)   # (child processes leave small traces of their own)
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gemm_idx = [i for i, r in enumerate(rows) if "gemm_f32_" in r[2]]
per_step = len(gemm_idx) // (warmup + steps)
start = gemm_idx[warmup * per_step]
region = rows[start:]
agg = collections.defaultdict(list)
for s, e, k in region:
    agg[k].append(e - s)
tot = sum(sum(v) for v in agg.values())
span = region[-1][1] - region[0][0]
with open(out, "w") as fh:
    fh.write("# timed region: %d steps, %d GEMM launches/step, span %.3f ms/step, kernel time %.3f ms/step\n"
             % (steps, per_step, span / steps / 1e6, tot / steps / 1e6))
    fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        fh.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (k.replace('"', "'"), len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
print(open(out).read()[:3000])
