"""Determinism soak: the same batch of proofs, proved again and again, must give the same bytes every time (a race in
the pipeline shows up as a differing proof long before it shows up as a wrong one), and sampled proofs must verify.
Usage: python scripts/soak.py [repeats_small [repeats_big]]   (defaults 100 40)"""
import faulthandler
import sys
import time

if __import__("os").environ.get("SOAK_WATCHDOG"):
    faulthandler.dump_traceback_later(int(__import__("os").environ["SOAK_WATCHDOG"]), exit=True)

sys.path.insert(0, ".")
import torch

import bench

rs_small = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs_big = int(sys.argv[2]) if len(sys.argv) > 2 else 40
z = bench.load_pkg().Zkmi()
ctx = z.context(0)
bad = 0
import os

CASES = ((13, 200, rs_small), (14, 130, rs_small), (16, 40, rs_small // 2), (18, 12, rs_big), (20, 12, rs_big))
if os.environ.get("SOAK_ODD"):  # the sizes in between (other group sizes, two partitions per proof, two-proof groups)
    CASES = ((15, 70, rs_small // 2), (17, 20, rs_big), (19, 7, rs_big), (21, 4, max(2, rs_big // 4)))
for lg, count, reps in CASES:
    r1, wits = bench.relation_and_witness(z, "poseidon", lg, [lg, lg + 100, lg + 200])
    rng = bench.SplitMix64(lg)
    pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
    d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
    idx = [i % 3 for i in range(count)]
    rs = [rng.fr_bytes() for _ in range(count)]
    ss = [rng.fr_bytes() for _ in range(count)]
    torch.cuda.synchronize()
    ref = ctx.groth16_prove_batch_dev(pk, [d[j].data_ptr() for j in idx], rs, ss)
    ok = all(z.groth16_verify(vk, wits[idx[i]][32 : 32 * r1.n_pub], ref[i]) for i in (0, count // 2, count - 1))
    t0 = time.time()
    diff = 0
    for rep in range(reps):
        if os.environ.get("SOAK_VERBOSE"):
            print(f"  2^{lg} rep {rep}", file=sys.stderr, flush=True)
        n = count if rep % 3 else count - (rep % 7)  # also partial last groups
        got = ctx.groth16_prove_batch_dev(pk, [d[j].data_ptr() for j in idx[:n]], rs[:n], ss[:n])
        diff += sum(1 for a, b in zip(got, ref) if a != b)
        if rep % 10 == 0:  # single-proof entry point in between
            i = rep % count
            diff += ctx.groth16_prove_dev(pk, d[idx[i]].data_ptr(), rs[i], ss[i]) != ref[i]
    print(f"2^{lg}: {reps} x {count} proofs in {time.time() - t0:.1f} s, sampled proofs verify: {ok}, differing proofs: {diff}", flush=True)
    bad += diff + (0 if ok else 1)
    pk.free()
    r1.free()
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
