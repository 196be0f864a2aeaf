#!/usr/bin/env python3
"""The two workloads whose time is made in the heavy-bucket path, for tracing and timing:
    python scripts/heavy_workloads.py msm  [log_n = 20] [reps = 4]    one G1 MSM over witness-like scalars (40 % zero,
                                                                      20 % one, 10 % below 2^16, 30 % uniform), plain and prepared bases
    python scripts/heavy_workloads.py bits [log_n = 20] [proofs = 8]  a relation of bit constraints (one heavy bucket per query)
Prints wall-clock medians; under `rocprofv3 --kernel-trace` the trace shows where the heavy-bucket launches sit."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402


def witness_like(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    uni = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    uni[:, 31] &= 0x3F
    kind = torch.rand(n, device="cuda", generator=g)
    mix = uni.clone()
    mix[(kind >= 0.6) & (kind < 0.7), 2:] = 0
    ones = (kind >= 0.4) & (kind < 0.6)
    mix[ones] = 0
    mix[ones, 0] = 1
    mix[kind < 0.4] = 0
    return uni, mix


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "msm"
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else (4 if mode == "msm" else 8)
    z = bench.load_pkg().Zkmi()
    ctx = z.context(0)
    if mode == "msm":
        n = 1 << lg
        uni, mix = witness_like(n, 0x5A4B)
        b = ctx.bases_g1_synthetic(n)
        for prepared in (False, True):
            if prepared:
                b.prepare()
            for name, sc in (("uniform", uni), ("witness_like", mix)):
                ctx.msm_g1_dev(sc.data_ptr(), n, b)
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    ctx.msm_g1_dev(sc.data_ptr(), n, b)
                    ts.append(time.perf_counter() - t0)
                print("G1 MSM 2^%d %-12s %-8s: %.3f ms (median of %d)" % (lg, name, "prepared" if prepared else "plain", 1e3 * sorted(ts)[len(ts) // 2], reps), flush=True)
        b.free()
    else:
        R = bench.R_MOD
        one = (1).to_bytes(32, "little")
        n_pub = 2
        nbits = (1 << lg) - n_pub - 8
        rnd = random.Random(lg)
        bits = [1 if rnd.random() < 0.6 else 0 for _ in range(nbits)]
        zv = [1, bits[0]] + bits
        cols = list(range(n_pub, n_pub + nbits))
        rp = list(range(nbits + 2))
        mats = [(rp, cols + [n_pub], one * (nbits + 1)), (rp, cols + [0], one * (nbits + 1)), (rp, cols + [1], one * (nbits + 1))]
        r1 = z.r1cs_create(n_pub + nbits, n_pub, mats)
        wit = b"".join(v.to_bytes(32, "little") for v in zv)
        prng = bench.SplitMix64(lg)
        pk, vk = ctx.groth16_setup(r1, b"".join(prng.fr_bytes() for _ in range(5)))
        d = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
        rs = [prng.fr_bytes() for _ in range(reps)]
        ss = [prng.fr_bytes() for _ in range(reps)]
        torch.cuda.synchronize()
        ctx.groth16_prove_batch_dev(pk, [d.data_ptr()] * 4, rs[:4], ss[:4])  # the fold decision settles here
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            proofs = ctx.groth16_prove_batch_dev(pk, [d.data_ptr()] * reps, rs, ss)
            best = min(best, time.perf_counter() - t0)
        ok = z.groth16_verify(vk, wit[32: 32 * n_pub], proofs[-1])
        t0 = time.perf_counter()
        ctx.groth16_prove_dev(pk, d.data_ptr(), rs[0], ss[0])
        one_ms = 1e3 * (time.perf_counter() - t0)
        print("bits relation 2^%d (%d ones): %.1f proofs/s, %.3f ms per proof, one proof alone %.2f ms, verified %s; schedule state %s"
              % (lg, sum(bits), reps / best, 1e3 * best / reps, one_ms, ok, pk.schedule_state()), flush=True)
        pk.free()
    ctx.close()


if __name__ == "__main__":
    main()
