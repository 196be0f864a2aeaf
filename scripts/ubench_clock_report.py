"""Join the plain run of scripts/ubench_clock (clock, power, time per body) with the PMC run (SQ_INSTS_VALU per dispatch) of
the same binary: wave-instructions per body-op, issue rate, share of the 4-cycle issue slots used at the measured clock.
Usage: python scripts/ubench_clock_report.py OUTDIR W"""
import collections
import csv
import glob
import re
import sys

out, w = sys.argv[1], sys.argv[2]
NAMES = ["sqr", "mul", "madd", "madd_lds", "add", "mul+2add", "mul+8add", "mul+32add", "mad_zero", "mad_random"]
OPS = {"add": 8.0}
# static multiply-add share of each body's loop (v_mad_{i,u}64 + v_mul_lo of all VALU instructions, from the disassembly)
plain = {}
for line in open(f"{out}/plain_w{w}.txt"):
    m = re.match(r"(\S+)\s+W=(\d)\s+iters\s+(\d+)\s+([\d.]+) ms\s+sclk ([\d.]+) GHz\s+([\d.]+) G body-ops/s\s+([\d.]+) ns/op/wave\s+power\s+(\d+) W", line)
    if m:
        plain[m.group(1)] = dict(iters=int(m.group(3)), ms=float(m.group(4)), ghz=float(m.group(5)), gops=float(m.group(6)), power=float(m.group(8)))
# PMC run: the LAST dispatch of every body is the measured one (calibration launches precede it)
tr = glob.glob(f"{out}/pmc_w{w}/**/*_kernel_trace.csv", recursive=True)
cc = glob.glob(f"{out}/pmc_w{w}/**/*_counter_collection.csv", recursive=True)
insts = collections.defaultdict(dict)
if cc:
    rows = list(csv.DictReader(open(cc[0])))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name = {}
    for r in rows:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    for d in sorted(per):
        m = re.search(r"k_body<(\d+), (\d+)>", name[d])
        if m:
            insts[NAMES[int(m.group(1))]] = per[d]  # later dispatches overwrite earlier ones
piters = {}
for line in open(f"{out}/pmc_w{w}.txt"):
    m = re.match(r"(\S+)\s+W=(\d)\s+iters\s+(\d+)", line)
    if m:
        piters[m.group(1)] = int(m.group(3))
print(f"{'body':<11} {'sclk GHz':>8} {'power W':>8} {'instr/op':>9} {'G instr/s':>10} {'slots used':>10}   (W = {w} waves per SIMD; slots = 1024 SIMDs x sclk / 4)")
for n in NAMES:
    if n not in plain:
        continue
    p = plain[n]
    ipo = rate = frac = float("nan")
    if n in insts and n in piters and "SQ_INSTS_VALU" in insts[n]:
        waves = 1024 * int(w)
        ipo = insts[n]["SQ_INSTS_VALU"] / (waves * piters[n] * OPS.get(n, 1.0))
        rate = ipo * p["gops"] / 64.0  # wave-instructions per second (gops counts lanes)
        frac = rate / (1024 * p["ghz"] / 4.0)
    print(f"{n:<11} {p['ghz']:>8.3f} {p['power']:>8.0f} {ipo:>9.0f} {rate:>10.1f} {frac:>10.2f}")
