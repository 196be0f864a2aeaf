"""Per accumulation-kernel launch of the rocprofv3 runs under DIR/alone_*: duration, clock (GRBM_GUI_ACTIVE / 8 / time),
resident waves per SIMD (4 x SQ_WAVE_CYCLES / (time x clock x 1024)), VALU issue rate."""
import collections
import csv
import glob
import os
import sys

for d in sorted(glob.glob(os.path.join(sys.argv[1], "alone_*/"))):
    tr = glob.glob(d + "**/*_kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "**/*_counter_collection.csv", recursive=True)
    if not tr or not cc:
        continue
    dur = {}
    for r in csv.DictReader(open(tr[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc[0])):
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = collections.defaultdict(list)
    for disp, (ns, name) in dur.items():
        if "k_accum" in name and "heavy" not in name and "redo" not in name and disp in agg:
            c = agg[disp]
            clk = c["GRBM_GUI_ACTIVE"] / 8 / (ns * 1e-9)
            occ = c["SQ_WAVE_CYCLES"] * 4 / (ns * 1e-9 * clk * 1024)
            rows[name.split("(")[0].split("::")[-1][:40]].append((ns / 1e6, clk / 1e9, occ, c["SQ_INSTS_VALU"] / ns, c["SQ_WAVES"]))
    for k, v in rows.items():
        n = len(v)
        print("%-22s %-40s ms %.3f  clk %.2f GHz  waves/SIMD %.2f  %.0f G instr/s  waves %d" % (
            os.path.basename(d.rstrip("/")), k, sum(x[0] for x in v) / n, sum(x[1] for x in v) / n, sum(x[2] for x in v) / n,
            sum(x[3] for x in v) / n, v[0][4]))
