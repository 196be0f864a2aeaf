"""rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU [...] serialises the dispatches, so
every kernel of the run is measured alone on the chip: duration, clock, resident waves per SIMD, issue rate.
Usage: python scripts/kernel_alone_report.py DIR [substring ...]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
want = sys.argv[2:]
tr = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
cc = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(tr)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cc)):
    agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = collections.defaultdict(list)
for disp, (ns, name, g, w) in dur.items():
    if disp not in agg or (want and not any(s in name for s in want)):
        continue
    c = agg[disp]
    clk = c["GRBM_GUI_ACTIVE"] / 8 / (ns * 1e-9)
    occ = c["SQ_WAVE_CYCLES"] * 4 / (ns * 1e-9 * clk * 1024) if clk else 0
    n = re.sub(r"^void ", "", name).replace("zkmi::", "").replace("(anonymous namespace)::", "").split("(")[0]
    n = re.sub(r"Fp28<(\w+)28Params\s*>", r"\g<1>28", n)
    rows[(n[:44], g, w)].append((ns / 1e6, clk / 1e9, occ, c["SQ_INSTS_VALU"] / ns, c["SQ_INSTS_VALU"]))
print("%-44s %9s %4s %4s %8s %5s %5s %6s %9s" % ("kernel", "grid", "wg", "n", "ms", "GHz", "w/SIMD", "Ginst/s", "insts"))
for (n, g, w), v in sorted(rows.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    k = len(v)
    print("%-44s %9s %4s %4d %8.3f %5.2f %5.2f %6.0f %9.3g" % (n, g, w, k, sum(x[0] for x in v) / k, sum(x[1] for x in v) / k,
                                                        sum(x[2] for x in v) / k, sum(x[3] for x in v) / k, sum(x[4] for x in v) / k))
