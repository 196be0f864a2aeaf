"""Per-phase HIP-event times (sums over streams: they overlap) of a batch of proofs at N = 2^lg.
Usage: python scripts/phase_breakdown.py LG [PROOFS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pkg = bench.load_pkg()
z = pkg.Zkmi(os.environ.get("ZKMI_LIB"))
ctx = z.context(0)
r1, wits = bench.relation_and_witness(z, "poseidon", lg, [1, 2])
rng = bench.SplitMix64(lg)
pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
rs = [rng.fr_bytes() for _ in range(count)]
ss = [rng.fr_bytes() for _ in range(count)]
torch.cuda.synchronize()
ptrs = [d[i % 2].data_ptr() for i in range(count)]
ctx.groth16_prove_batch_dev(pk, ptrs[:3], rs[:3], ss[:3])
ctx.sync()
t0 = time.perf_counter()
ctx.groth16_prove_batch_dev(pk, ptrs, rs, ss)
ctx.sync()
dt = time.perf_counter() - t0
ctx.prof_enable(True)
ctx.prof_reset()
p = ctx.groth16_prove_batch_dev(pk, ptrs, rs, ss)
ctx.sync()
ctx.prof_enable(False)
ph = {k: round(ctx.prof_get(k)[0] / count, 3) for k in pkg.PHASES}
print(f"2^{lg}: {count / dt:.2f} proofs/s, {1e3 * dt / count:.2f} ms per proof; phase ms per proof (event sums, overlapping): {ph}; verified",
      z.groth16_verify(vk, wits[(count - 1) % 2][32: 32 * r1.n_pub], p[-1]), flush=True)
