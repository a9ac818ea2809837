"""Effective shader clock and matrix-pipe utilisation per kernel from a
`rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace` run:
  clock    = GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / 8 / dispatch duration
  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)      (clock-independent)
usage: pmc_clock.py <dir> [name-filter] [list]"""
import collections, csv, glob, sys
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm_f32_stream"
dur = {}
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or flt not in r["Kernel_Name"]:
            continue
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        elif r["Dispatch_Id"] in dur:
            d = dur[r["Dispatch_Id"]][0]
        else:
            continue
        if d > 200000:      # launches long enough for the ratio to mean something
            agg[r["Kernel_Name"][:110] + " grid=" + r["Grid_Size"]].append(float(r["Counter_Value"]) / 8.0 / d)
mfma = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"] and r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            mfma[(r["Dispatch_Id"], r["Kernel_Name"][:110] + " grid=" + r["Grid_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
util = collections.defaultdict(list)
for (d, k), c in mfma.items():
    if len(c) == 2 and c["GRBM_GUI_ACTIVE"] > 8 * 400000:
        util[k].append(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
if util:
    print("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs), launches >= ~0.2 ms")
    allv = []
    for k in sorted(util):
        v = sorted(util[k]); allv += v
        print("%6.1f %% median  %6.1f min  %6.1f max  n=%-4d %s" % (100 * v[len(v) // 2], 100 * v[0], 100 * v[-1], len(v), k))
    print("%6.1f %% mean over all %d launches" % (100 * sum(allv) / len(allv), len(allv)))
print("effective clock (GHz) = GRBM_GUI_ACTIVE / 8 / duration, launches >= 0.2 ms")
for k in sorted(agg):
    v = sorted(agg[k])
    print("%6.3f median  %6.3f min  %6.3f max  n=%-4d %s" % (v[len(v) // 2], v[0], v[-1], len(v), k))

if len(sys.argv) > 3 and sys.argv[3] == "list":
    # every dispatch >= 30 us of the last third of the run, in time order: name, duration, effective clock
    import re
    rows = []
    for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            else:
                continue
            rows.append((s, e - s, float(r["Counter_Value"]), r["Kernel_Name"], r["Grid_Size"]))
    rows.sort()
    t0 = rows[0][0] + (rows[-1][0] - rows[0][0]) * 2 // 3
    for s, d, v, k, g in rows:
        if s < t0 or d < 30000:
            continue
        m = re.search(r"(\w+)<ait_gemm::Cfg<([\d, ]+)>, (\w+), (\w+), (\d)", k)
        name = ("gemm %s/%s epi%s" % (m.group(3)[0], m.group(4)[0], m.group(5))) if m else re.sub(r"^void |\(anonymous namespace\)::|\(.*", "", k)[:60]
        print("%9.3f ms  %8.1f us  %5.2f GHz  %s" % ((s - t0) / 1e6, d / 1e3, v / 8.0 / d, name))
