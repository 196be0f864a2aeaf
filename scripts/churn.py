"""Key-lifecycle churn: the stress the round-2 verdict asked for after an unexplained SIGSEGV in the prover.

Random interleaving, on ONE context (plus a short-lived second one), of everything that allocates, re-reserves or
frees workspaces: trusted setups of sizes 2^13 .. 2^max in shuffled order (big -> small transitions forced), forced
group sizes 64/32/16/8/1 (zkmi_ctx_set_group_size, per key), batch proofs with partial last groups, single proofs from
device and host witnesses, the unsatisfied-witness error path, generic MSMs (plain, prepared, >= 2^21 terms so the
sort buffers grow past what a grouped key reserved), NTTs, BN254 calls, key frees in random order.  Every proof is
compared byte for byte with the first proof ever made from the same (size, witness, r, s) -- keys are re-created
from the same toxic waste, so a lifetime or race bug shows up as a differing proof, an error code or a crash -- and
a sample is verified by pairing.

  python scripts/churn.py [--ops 10000] [--seconds 0] [--seed 1] [--max-log-n 20] [--verbose]
Run with ZKMI_BACKTRACE=1 for a native stack on a fatal signal.  Exit code 0 = clean.
"""
import argparse
import faulthandler
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()

import torch  # noqa: E402

import bench  # noqa: E402  (relation_and_witness, SplitMix64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", type=int, default=10000)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop after this many seconds (0 = run all ops)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--min-log-n", type=int, default=13)
    ap.add_argument("--max-log-n", type=int, default=20)
    ap.add_argument("--max-keys", type=int, default=3)
    ap.add_argument("--watchdog", type=int, default=0, help="dump tracebacks and exit if the run takes longer (s)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    if args.watchdog:
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    rnd = random.Random(args.seed)
    pkg = bench.load_pkg()
    z = pkg.Zkmi()
    ctx = z.context(0)
    build_v, run_v = z.hip_versions()
    print(f"churn: HIP build {build_v} runtime {run_v} seed {args.seed} ops {args.ops}", flush=True)

    sizes = list(range(args.min_log_n, args.max_log_n + 1))
    relations = {}   # lg -> (r1cs, [host witness bytes], [device tensors], [pinned host tensors])
    expected = {}    # (lg, wit index, rs index) -> proof bytes of the first time
    keys = []        # dicts: lg, pk, vk, group
    rs_pool = {}
    stats = {"setup": 0, "free": 0, "batch": 0, "single": 0, "host": 0, "unsat": 0, "msm": 0, "ntt": 0, "ctx2": 0,
             "proofs": 0, "verified": 0, "big_to_small": 0}
    bad = []

    def rel(lg):
        if lg not in relations:
            r1, wits = bench.relation_and_witness(z, "poseidon", lg, [lg * 7 + 1, lg * 7 + 2])
            dev = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
            pin = [torch.frombuffer(bytearray(w), dtype=torch.uint8).pin_memory() for w in wits]
            torch.cuda.synchronize()
            rng = bench.SplitMix64(0xC0DE + lg)
            rs_pool[lg] = [(rng.fr_bytes(), rng.fr_bytes()) for _ in range(3)]
            relations[lg] = (r1, wits, dev, pin)
        return relations[lg]

    def toxic(lg):
        rng = bench.SplitMix64(0x70C51C + lg)
        return b"".join(rng.fr_bytes() for _ in range(5))

    def check(lg, wi, ri, proof, what):
        k = (lg, wi, ri)
        if k not in expected:
            expected[k] = proof
            r1, wits, _, _ = rel(lg)
            key = next(kk for kk in keys if kk["lg"] == lg)
            if not z.groth16_verify(key["vk"], wits[wi][32: 32 * r1.n_pub], proof):
                bad.append(f"{what}: first proof of {k} does not verify")
            stats["verified"] += 1
        elif expected[k] != proof:
            bad.append(f"{what}: proof of {k} differs from the first one")
        stats["proofs"] += 1

    last_setup_lg = [None]

    def op_setup(force_lg=None):
        if len(keys) >= args.max_keys:
            op_free()
        have = {k["lg"] for k in keys}
        cand = [s for s in sizes if s not in have]
        if not cand:
            return
        # favour small sizes (cheap), but keep the big ones coming; after a big key prefer a much smaller one
        if force_lg is not None and force_lg in cand:
            lg = force_lg
        elif last_setup_lg[0] is not None and last_setup_lg[0] >= 17 and rnd.random() < 0.7:
            lg = rnd.choice([s for s in cand if s <= 15] or cand)
        else:
            w = [3.0 if s <= 15 else 1.5 if s <= 17 else 0.5 for s in cand]
            lg = rnd.choices(cand, weights=w)[0]
        if last_setup_lg[0] is not None and last_setup_lg[0] > lg:
            stats["big_to_small"] += 1
        last_setup_lg[0] = lg
        r1, _, _, _ = rel(lg)
        g = rnd.choice([None, None, 64, 32, 16, 8, 1])
        ctx.set_group_size(0 if g is None else g)
        pk, vk = ctx.groth16_setup(r1, toxic(lg))
        ctx.set_group_size(0)
        keys.append({"lg": lg, "pk": pk, "vk": vk, "group": g})
        stats["setup"] += 1
        if args.verbose:
            print(f"  setup 2^{lg} group={g}", flush=True)

    def op_free():
        if not keys:
            return
        k = keys.pop(rnd.randrange(len(keys)))
        k["pk"].free()
        stats["free"] += 1
        if args.verbose:
            print(f"  free 2^{k['lg']}", flush=True)

    def pick_key():
        if not keys:
            op_setup()
        return rnd.choice(keys)

    def op_batch():
        k = pick_key()
        lg = k["lg"]
        _, _, dev, _ = rel(lg)
        cap = 200 if lg <= 14 else 70 if lg <= 16 else 12 if lg <= 18 else 5
        n = rnd.choice([1, 2, 3, rnd.randrange(1, cap + 1), cap])
        idx = [(rnd.randrange(2), rnd.randrange(3)) for _ in range(n)]
        got = ctx.groth16_prove_batch_dev(k["pk"], [dev[w].data_ptr() for w, _ in idx], [rs_pool[lg][r][0] for _, r in idx],
                                          [rs_pool[lg][r][1] for _, r in idx])
        for (w, r), p in zip(idx, got):
            check(lg, w, r, p, "batch")
        stats["batch"] += 1

    def op_single():
        k = pick_key()
        lg = k["lg"]
        _, _, dev, _ = rel(lg)
        w, r = rnd.randrange(2), rnd.randrange(3)
        check(lg, w, r, ctx.groth16_prove_dev(k["pk"], dev[w].data_ptr(), *rs_pool[lg][r]), "single")
        stats["single"] += 1

    def op_host():
        k = pick_key()
        lg = k["lg"]
        _, wits, _, pin = rel(lg)
        if rnd.random() < 0.5:
            w, r = rnd.randrange(2), rnd.randrange(3)
            check(lg, w, r, ctx.groth16_prove(k["pk"], wits[w], *rs_pool[lg][r]), "host single")
        else:
            n = rnd.randrange(1, 6)
            idx = [(rnd.randrange(2), rnd.randrange(3)) for _ in range(n)]
            got = ctx.groth16_prove_batch_host(k["pk"], [pin[w].data_ptr() for w, _ in idx], [rs_pool[lg][r][0] for _, r in idx],
                                               [rs_pool[lg][r][1] for _, r in idx])
            for (w, r), p in zip(idx, got):
                check(lg, w, r, p, "host batch")
        stats["host"] += 1

    def op_unsat():
        k = pick_key()
        lg = k["lg"]
        _, wits, dev, _ = rel(lg)
        flipped = bytearray(wits[0])
        flipped[32 * 4000] ^= 1
        d = torch.frombuffer(flipped, dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        ptrs = [dev[0].data_ptr(), d.data_ptr(), dev[1].data_ptr()]
        try:
            if rnd.random() < 0.5:
                ctx.groth16_prove_dev(k["pk"], d.data_ptr(), *rs_pool[lg][0])
            else:
                ctx.groth16_prove_batch_dev(k["pk"], ptrs, [rs_pool[lg][0][0]] * 3, [rs_pool[lg][0][1]] * 3)
            bad.append("unsatisfied assignment was proved")
        except pkg.ZkmiError as e:
            if e.code != -8:
                bad.append(f"unsatisfied assignment: code {e.code}")
        del d
        stats["unsat"] += 1

    bases_cache = {}

    def op_msm():
        # generic MSM entry points on the same context: plain (windowed) and prepared (shared buckets); sizes up to
        # 2^22 so that MsmSort::reserve grows past what the grouped keys reserved
        lg = rnd.choice([8, 10, 12, 14, 16, 18, 20, 21, 22] if args.max_log_n >= 20 else [8, 10, 12, 14, 16, 18])
        n = (1 << lg) - rnd.choice([0, 0, 1, 37])
        key = (lg, n)
        if key not in bases_cache:
            if len(bases_cache) >= 4:
                _, (b0, *_rest) = bases_cache.popitem()
                b0.free()
            b = ctx.bases_g1_synthetic(n)
            sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(lg))
            sc[:, 31] &= 0x3F
            torch.cuda.synchronize()
            want = ctx.msm_g1_dev(sc.data_ptr(), n, b)
            bases_cache[key] = (b, sc, want, [False])
        b, sc, want, prepared = bases_cache[key]
        if not prepared[0] and lg <= 21 and rnd.random() < 0.3:
            b.prepare()
            prepared[0] = True
        got = ctx.msm_g1_dev(sc.data_ptr(), n, b)
        if got != want:
            bad.append(f"msm_g1 n={n} differs from its first result (prepared={prepared[0]})")
        stats["msm"] += 1

    def op_ntt():
        lg = rnd.choice([4, 9, 10, 11, 14, 16, 18, 20])
        n = 1 << lg
        a = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
        a[:, 31] &= 0x3F
        ref = a.clone()
        torch.cuda.synchronize()
        cos = rnd.random() < 0.5
        ctx.ntt_dev(a.data_ptr(), lg, coset=cos)
        ctx.ntt_dev(a.data_ptr(), lg, inverse=True, coset=cos)
        ctx.sync()
        if not torch.equal(a, ref):
            bad.append(f"ntt round trip 2^{lg} coset={cos}")
        stats["ntt"] += 1

    def op_ctx2():
        c2 = z.context(0)
        b = c2.bases_g1_synthetic(1 << 10)
        sc = torch.randint(0, 256, (1 << 10, 32), dtype=torch.uint8, device="cuda")
        sc[:, 31] &= 0x3F
        torch.cuda.synchronize()
        x = c2.msm_g1_dev(sc.data_ptr(), 1 << 10, b)
        b2 = ctx.bases_g1_synthetic(1 << 10)
        if ctx.msm_g1_dev(sc.data_ptr(), 1 << 10, b2) != x:
            bad.append("second context disagrees with the first")
        b.free()
        b2.free()
        c2.close()
        stats["ctx2"] += 1

    ops = [(op_setup, 8), (op_free, 5), (op_batch, 30), (op_single, 22), (op_host, 8), (op_unsat, 4), (op_msm, 12), (op_ntt, 6), (op_ctx2, 1)]
    fns, weights = zip(*ops)
    t0 = time.time()
    done = 0
    # the sequence of the round-2 crash first: a 2^16 key proved and freed, then a 2^15 key and a single proof
    if args.min_log_n <= 15 and args.max_log_n >= 16:
        op_setup(16)
        op_batch()
        op_free()
        op_setup(15)
        op_single()
    for done in range(1, args.ops + 1):
        fn = rnd.choices(fns, weights=weights)[0]
        try:
            fn()
        except pkg.ZkmiError as e:
            bad.append(f"{fn.__name__}: {e}")
        if bad:
            break
        if args.seconds and time.time() - t0 > args.seconds:
            break
        if done % 500 == 0:
            print(f"  {done} ops, {time.time() - t0:.0f} s, {stats}", flush=True)
    for k in keys:
        k["pk"].free()
    for b, *_ in bases_cache.values():
        b.free()
    for r1, *_ in relations.values():
        r1.free()
    ctx.close()
    print(f"churn: {done} ops in {time.time() - t0:.1f} s: {stats}")
    if bad:
        print("CHURN FAILED:", *bad, sep="\n  ")
        return 1
    print("CHURN OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
