"""Aggregate proof rate at small domain sizes with K independent provers (ctx + key each, one host thread
each) sharing ONE GPU: small proofs cannot fill the chip from a single pipeline (a 2^14-term bucket pass is
128 waves), several pipelines side by side can.  Usage: python scripts/small_n_concurrency.py [log_n=14] [steps=200]"""
import sys
import threading
import time

sys.path.insert(0, ".")
import torch

import bench
from zkmi_loader import load_pkg

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pkg = load_pkg()
z = pkg.Zkmi()
torch.cuda.init()


def prover(k, out, start, go):
    ctx = z.context(0)
    r1, wits = bench.relation_and_witness(z, "poseidon", lg, [100 + 2 * k, 101 + 2 * k])
    rng = bench.SplitMix64(7 + k)
    pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
    d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
    torch.cuda.synchronize()
    rs = [rng.fr_bytes() for _ in range(steps)]
    ptrs = [d[i % 2].data_ptr() for i in range(steps)]
    ctx.groth16_prove_batch_dev(pk, ptrs[:8], rs[:8], rs[:8])  # warm-up
    start.wait()
    go.wait()
    proofs = ctx.groth16_prove_batch_dev(pk, ptrs, rs, rs)
    out[k] = z.groth16_verify(vk, wits[(steps - 1) % 2][32 : 32 * r1.n_pub], proofs[-1])
    pk.free()
    ctx.close()


for K in (1, 2, 4, 8):
    out = [None] * K
    start, go = threading.Barrier(K + 1), threading.Event()
    th = [threading.Thread(target=prover, args=(k, out, start, go)) for k in range(K)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    go.set()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print(f"N=2^{lg}: {K} prover(s) on one GPU: {K * steps / dt:.0f} proofs/s aggregate ({dt / steps * 1e3:.2f} ms per step), verified: {all(out)}")
