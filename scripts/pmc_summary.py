"""Summarise rocprofv3 --pmc passes (counter_collection.csv files) into one JSON: per kernel, the
average counter value per dispatch.  Usage: python scripts/pmc_summary.py OUT.json PROOFS DIR [DIR ...]
(each DIR is the -d directory of one rocprofv3 --pmc pass; PROOFS = warm-up + timed proofs each pass ran,
recorded in a "__meta__" row so that per-proof totals can be derived)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("zkmi::", "")
    name = name.split("(")[0]
    name = re.sub(r"Fp28<(\w+)28Params\s*>", r"\g<1>28", name)
    name = re.sub(r"Fq2T<Fq28\s*>", "Fq2_28", name)
    return name.strip()


def main():
    out_path, proofs, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            # one row per (dispatch, counter); a counter may appear once per dimension instance: sum those
            per_dispatch = collections.defaultdict(float)
            names = {}
            span = {}
            for r in csv.DictReader(open(path)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    span[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            for (disp, counter), v in per_dispatch.items():
                agg[names[disp]][counter].append(v)
                # the clock the chip held under this dispatch (counter passes serialise the kernels: every dispatch ran alone):
                # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles (MI355X_MICROARCH.md, DVFS give-back)
                if counter == "GRBM_GUI_ACTIVE" and span.get(disp, 0) > 0:
                    agg[names[disp]]["alone_ns"].append(span[disp])
                    agg[names[disp]]["held_clock_ghz"].append(v / 8.0 / span[disp])
    rows = []
    for kernel, counters in agg.items():
        row = {"kernel": kernel}
        for c, vals in sorted(counters.items()):
            row[c + "_avg_per_dispatch"] = sum(vals) / len(vals)
            row[c + "_dispatches"] = len(vals)
        rows.append(row)
    rows.sort(key=lambda r: -r.get("SQ_INSTS_VALU_avg_per_dispatch", 0) * r.get("SQ_INSTS_VALU_dispatches", 0))
    # the digest of the kernel sources the profiled library was built from (bench.py prints the same digest of the
    # sources it runs: a summary older than the library shows as a mismatch in roofline.traffic_source)
    import bench

    rows.insert(0, {"kernel": "__meta__", "proofs": proofs, "passes": dirs, "csrc_sha256_16": bench.csrc_digest()})
    json.dump(rows, open(out_path, "w"), indent=1)
    print("wrote", out_path, len(rows), "kernels")


if __name__ == "__main__":
    main()
