"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel launches / total / average for
the kernels of libait_hip.so.  usage: pmc_summary.py <dir> <out.csv>"""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"\(anonymous namespace\)::(\w+)|ait_gemm::(\w+)", k)
        if not m or k.startswith("void at::") or "at::native" in k or "rocprim" in k:
            continue
        name = m.group(1) or m.group(2)
        if name.startswith("gemm_f32_"):
            name += "[grid=%s]" % r["Grid_Size"]
        a = agg[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[2], "w") as out:
    out.write("kernel,counter,launches,total,avg_per_launch\n")
    for k in sorted(agg):
        for c, (v, n) in sorted(agg[k].items()):
            out.write("%s,%s,%d,%.0f,%.1f\n" % (k, c, n, v, v / n))
print(open(sys.argv[2]).read())
