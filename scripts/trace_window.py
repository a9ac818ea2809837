"""The kernels of one step of a rocprofv3 --kernel-trace csv of bench.py between two kernel-name patterns (first match of
<from> ... first later match of <to>), with start offsets, durations and gaps: what the GPU does around a host sync.
usage: trace_window.py <dir> <warmup> <steps> <from-pattern> <to-pattern>"""
import csv, glob, os, re, sys
d, warmup, steps, p0, p1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f)))
gi = [i for i, r in enumerate(rows) if "gemm_f32_" in r[2]]
per = len(gi) // (warmup + steps)
lo, hi = gi[(warmup + steps - 2) * per], gi[(warmup + steps - 1) * per]
region = rows[lo:hi]
i0 = next(i for i, r in enumerate(region) if re.search(p0, r[2]))
i1 = next(i for i in range(i0 + 1, len(region)) if re.search(p1, region[i][2]))
t0 = region[max(0, i0 - 6)][0]
short = lambda k: re.sub(r"^void |\(anonymous namespace\)::|at::native::|\(.*", "", k)[:90]
prev_end = None
for s, e, k, q in region[max(0, i0 - 6):i1 + 4]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  +%7.1f us  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q[-2:], short(k)))
    prev_end = max(prev_end or 0, e)
