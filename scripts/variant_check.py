"""One digest over results that must not depend on any A/B switch (ZKMI_NTT_RB, ZKMI_SORT_FINE, ZKMI_HEAVY_ON, ZKMI_ACCUM, ...):
NTTs of several sizes (one-, two- and three-pass plans) in all four modes, a prepared and a plain G1 MSM with uniform and
witness-like scalars, and Groth16 proofs at 2^13 (grouped) and 2^17 (two bucket partitions per proof: the sizes where the
record / fine-partition sorts run).  The switches are read once per process, so tests/test_gpu_sizes.py runs this script in
a child process per variant and compares the printed digests.  Usage: python scripts/variant_check.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

z = bench.load_pkg().Zkmi()
ctx = z.context(0)
h = hashlib.sha256()
rng = np.random.default_rng(2024)
for lg in (9, 12, 16, 20, 21):
    a = rng.integers(0, 256, size=(1 << lg, 32), dtype=np.uint8)
    a[:, 31] &= 0x3F
    x = a.tobytes()
    for inverse in (False, True):
        for coset in (False, True):
            h.update(ctx.ntt(x, lg, inverse=inverse, coset=coset))
n = (1 << 18) - 5
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
sc[:, 31] &= 0x3F
wl = sc.copy()
kind = rng.random(n)
wl[kind < 0.5] = 0
wl[(kind >= 0.3) & (kind < 0.5), 0] = 1
b = ctx.bases_g1_synthetic(n)
h.update(ctx.msm_g1(sc.tobytes(), b))
h.update(ctx.msm_g1(wl.tobytes(), b))
b.prepare()
h.update(ctx.msm_g1(sc.tobytes(), b))
h.update(ctx.msm_g1(wl.tobytes(), b))
b.free()
for lg, count in ((13, 70), (17, 9)):
    r1, wits = bench.relation_and_witness(z, "poseidon", lg, [lg, lg + 1])
    prng = bench.SplitMix64(lg)
    pk, vk = ctx.groth16_setup(r1, b"".join(prng.fr_bytes() for _ in range(5)))
    d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
    rs = [prng.fr_bytes() for _ in range(count)]
    ss = [prng.fr_bytes() for _ in range(count)]
    torch.cuda.synchronize()
    proofs = ctx.groth16_prove_batch_dev(pk, [d[i % 2].data_ptr() for i in range(count)], rs, ss)
    assert z.groth16_verify(vk, wits[0][32: 32 * r1.n_pub], proofs[0])
    assert ctx.groth16_prove_dev(pk, d[1].data_ptr(), rs[1], ss[1]) == proofs[1]
    for p in proofs:
        h.update(p)
    pk.free()
    r1.free()
ctx.close()
print("VARIANT_DIGEST", h.hexdigest())
