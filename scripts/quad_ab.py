#!/usr/bin/env python3
"""One line of figures for an A/B of the reduction-side kernels (csrc/msm_quad.hpp; A/B library: ZKMI_QUAD = mask of the
latency contexts, ZKMI_QUAD_BATCH = mask inside the batch prover): single-proof latency at 2^14 and 2^20, group rate at
2^14, batch rate at 2^20 (16 proofs), one G1 MSM of 2^20 terms (uniform / witness-like, plain / prepared bases).  Usage: ZKMI_LIB=.../libzkmi_exp.so ZKMI_QUAD=.. python scripts/quad_ab.py [what ...]"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402


def med(f, reps):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        ts.append(1e3 * (time.perf_counter() - t0))
    return round(statistics.median(ts), 3)


def main():
    what = set(sys.argv[1:]) or {"single14", "single20", "group14", "batch20", "msm"}
    z = bench.load_pkg().Zkmi(os.environ.get("ZKMI_LIB"))
    ctx = z.context(0)
    out = {"QUAD": os.environ.get("ZKMI_QUAD"), "QUAD_BATCH": os.environ.get("ZKMI_QUAD_BATCH")}
    rng = bench.SplitMix64(3)
    for lg in (14, 20):
        if not ({"single%d" % lg, "group%d" % lg, "batch%d" % lg} & what):
            continue
        r1, wits = bench.relation_and_witness(z, "poseidon", lg, [1, 2])
        pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
        d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
        r, s = rng.fr_bytes(), rng.fr_bytes()
        if "single%d" % lg in what:
            for i in range(3):
                p = ctx.groth16_prove_dev(pk, d[i % 2].data_ptr(), r, s)
            ctx.sync()
            out["single_2p%d_ms" % lg] = med(lambda: ctx.groth16_prove_dev(pk, d[0].data_ptr(), r, s), 15 if lg == 14 else 7)
            assert z.groth16_verify(vk, wits[0][32 : 32 * r1.n_pub], ctx.groth16_prove_dev(pk, d[0].data_ptr(), r, s))
        n = 512 if lg == 14 else 16
        if ("group14" in what and lg == 14) or ("batch20" in what and lg == 20):
            ptrs = [d[i % 2].data_ptr() for i in range(n)]
            rs, ss = [r] * n, [s] * n
            ctx.groth16_prove_batch_dev(pk, ptrs, rs, ss)
            ms = med(lambda: ctx.groth16_prove_batch_dev(pk, ptrs, rs, ss), 3)
            out["%s_proofs_per_s" % ("group_2p14" if lg == 14 else "batch_2p20")] = round(n / ms * 1e3, 1)
        pk.free()
        r1.free()
        del d
        torch.cuda.empty_cache()
    if "msm" in what:
        import heavy_workloads as hw

        n = 1 << 20
        uni, mix = hw.witness_like(n, 0x5A4B)
        b = ctx.bases_g1_synthetic(n)
        for prepared in (False, True):
            if prepared:
                b.prepare()
            for name, sc in (("uniform", uni), ("witness_like", mix)):
                ctx.msm_g1_dev(sc.data_ptr(), n, b)
                out["msm_2p20_%s_%s_ms" % (name, "prepared" if prepared else "plain")] = med(lambda: ctx.msm_g1_dev(sc.data_ptr(), n, b), 7)
        b.free()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
