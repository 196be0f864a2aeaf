#!/usr/bin/env python3
"""Static opcode mix of the bucket-accumulation loop (k_accum_g1_nc<Fq28, 3, 1>) priced with MEASURED issue costs.

    python scripts/isa_mix_report.py [ubench_ops output]      (needs hipcc; no GPU)

Compiles zk-apps_amd/csrc/msm_g1.hip with -save-temps, takes the kernel's main loop (the blocks inside its outermost
backward branch around the largest basic block), counts VALU / SALU / memory opcodes, and -- given the output of
scripts/_bin/ubench_ops from the GPU box -- prices every VALU opcode at its measured SIMD-ticks per wave-instruction at
2 waves per SIMD (the rows whose ticks and event-timed chip rates agree with each other and with round 4's one-second runs:
4.19 ticks per multiply-add; the 3-wave rows of a 0.2 ms launch do not -- see the note the script prints).  The result is the issue ceiling of the kernel's ACTUAL mix, beside the flat 4-cycles-per-instruction
ceiling (614.4 G wave-instr/s at 2.4 GHz) bench.py quotes."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN4zkmi13k_accum_g1_ncINS_4Fp28INS_10Fq28ParamsEEELi3ELi1ELb0ELb0EEE"


def asm_text():
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-exceptions", "-save-temps=obj",
                               "-c", os.path.join(ROOT, "zk-apps_amd", "csrc", "msm_g1.hip"), "-o", os.path.join(d, "msm_g1.o")],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(os.path.join(d, "msm_g1-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def loop_blocks(text):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
    end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))
    blocks, cur = [], ("entry", [])
    for l in lines[start + 1: end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        s = l.split(";")[0].strip()
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
        elif s and not s.startswith((".", "//")):
            cur[1].append(s)
    blocks.append(cur)
    order = {name: i for i, (name, _) in enumerate(blocks)}
    biggest = max(range(len(blocks)), key=lambda i: len(blocks[i][1]))
    best = None
    for i, (name, ins) in enumerate(blocks):
        for s in ins:
            if s.startswith(("s_cbranch", "s_branch")):
                tgt = s.split()[-1]
                if tgt in order and order[tgt] <= biggest <= i and (best is None or i - order[tgt] > best[1] - best[0]):
                    best = (order[tgt], i)
    meta = [l.strip() for l in lines[end: end + 80] if "NumVgprs" in l or "ScratchSize" in l or "Occupancy" in l]
    return blocks[best[0]: best[1] + 1], meta


def main():
    blocks, meta = loop_blocks(asm_text())
    ops = collections.Counter()
    rare = collections.Counter()
    for name, ins in blocks:
        # the full zero test of P^2 (compare against k p, |k| <= 4) sits behind a one-limb test: executed for 2^-28 of the additions
        tgt = rare if sum(1 for s in ins if s.startswith("v_cmp_eq_u32")) > 50 else ops
        for s in ins:
            tgt[s.split()[0].replace("_e32", "").replace("_e64", "")] += 1
    total = sum(ops.values())
    print("kernel: k_accum_g1_nc<Fq28, 3, 1, false, false>;", "; ".join(meta))
    print("main loop: %d basic blocks, %d instructions on the common path (+ %d behind the one-limb zero test)" % (len(blocks), total, sum(rare.values())))
    cost = {}
    if len(sys.argv) > 1:
        for l in open(sys.argv[1]):
            m = re.match(r"OP (\S+)\s+waves/SIMD 2\s+SIMD-ticks/instr\s+([\d.]+)", l)
            if m:
                cost[m.group(1)] = float(m.group(2))
    if "v_add_co_u32+v_addc_co_u32" in cost:
        cost["v_add_co_u32"] = cost["v_addc_co_u32"] = cost["v_add_co_u32+v_addc_co_u32"]
    alias = {"v_lshlrev_b64": "v_lshrrev_b64", "v_ashrrev_i32": "v_ashrrev_i32", "v_lshrrev_b32": "v_lshlrev_b32", "v_cndmask_b32": "v_mov_b32",
             "v_mov_b64": "v_lshl_add_u64", "v_or_b32": "v_and_b32", "v_xor_b32": "v_and_b32", "v_bitop3_b32": "v_or3_b32", "v_cmp_lt_i32": "v_cmp_eq_u32",
             "v_cmp_ne_u32": "v_cmp_eq_u32", "v_cmp_gt_u32": "v_cmp_eq_u32", "v_cmp_lt_u32": "v_cmp_eq_u32", "v_add3_u32": "v_or3_b32",
             "v_subrev_u32": "v_sub_u32", "v_cmp_eq_u64": "v_cmp_eq_u32", "v_cmp_ne_u64": "v_cmp_eq_u32", "v_readfirstlane_b32": "v_mov_b32",
             "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32", "v_lshlrev_b32": "v_lshlrev_b32"}
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    nv = sum(valu.values())
    print("VALU %d (%.1f %%), scalar %d, memory/LDS/waits %d" % (nv, 100.0 * nv / total, sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith("s_wait")),
                                                             sum(v for k, v in ops.items() if not k.startswith(("v_", "s_")) or k.startswith("s_wait"))))
    print("%-22s %6s %7s %12s %10s" % ("opcode", "count", "share", "ticks/instr", "ticks"))
    ticks = unknown = 0.0
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
        c = cost.get(k, cost.get(alias.get(k, ""), None))
        if c is None:
            unknown += v
        print("%-22s %6d %6.1f%% %12s %10s" % (k, v, 100.0 * v / nv, "%.2f" % c if c else "?", "%.0f" % (c * v) if c else "?"))
        if c:
            ticks += c * v
    if cost:
        priced = nv - unknown
        mac = sum(v for k, v in valu.items() if k.startswith("v_mad_"))
        print("priced %d of %d VALU instructions: %.0f SIMD-ticks per addition = %.3f ticks per instruction on average" % (priced, nv, ticks, ticks / priced))
        print("multiply-adds: %d (%.1f %% of VALU) at %.2f ticks" % (mac, 100.0 * mac / nv, cost.get("v_mad_i64_i32", 0)))
        macc = cost.get("v_mad_i64_i32", 4.0)
        rel = ticks / priced / macc  # average cost of an instruction of this mix in multiply-add slots
        print("average cost of an instruction of this mix = %.3f multiply-add slots (a multiply-add = %.2f ticks = one 4-cycle issue slot:"
              " round 4's one-second runs measured 4.08 cycles at 3 waves per SIMD)" % (rel, macc))
        print("issue ceiling of THIS mix: 614.4 / %.3f = %.1f G wave-instr/s at 2.4 GHz (flat ceiling, every instruction one multiply-add slot: 614.4);"
              " at the ~2.0 GHz the chip holds under this kernel: %.1f" % (rel, 614.4 / rel, 512.0 / rel))
        print("note: the 3-waves-per-SIMD rows of ubench_ops (768-thread blocks, 0.2 ms launches) read 2.80 ticks per multiply-add with a LOWER event-timed"
              " chip rate than the 2-wave rows -- ticks and rate disagree there, so they are not used; relative costs are the same in both (simple VOP2 ops"
              " 0.55 of a multiply-add, 64-bit / VOP3 / v_mul_lo 1.05)")


if __name__ == "__main__":
    main()
