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
# the same slices under the plan of a 2^24-term MSM (20-bit windows: the two-level record sort of msm_sort.hip run_windowed_big;
# the witness-like slice puts 52 000 records into one fine partition: the round form of k_fpart_sort)
for s_ in (sc, wl):
    d_ = torch.from_numpy(s_).cuda()
    torch.cuda.synchronize()
    w_, nwin_, cb_ = ctx.msm_g1_windows_dev(d_.data_ptr(), n, b, 1 << 24)
    assert (nwin_, cb_) == (13, 20)
    h.update(w_)
b.prepare()
h.update(ctx.msm_g1(sc.tobytes(), b))
h.update(ctx.msm_g1(wl.tobytes(), b))
b.free()
# degenerate inputs (the pattern of tests/test_gpu_parity.py::test_msm_degenerate_bases_and_scalars) through the plain and the
# prepared G1 MSM: equal points with equal scalars (P + P), P + (-P), points at infinity, scalars 0, 1, r - 1
FQ = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FR = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def g1_neg(pt):  # 96 bytes x || y, little-endian canonical
    y = int.from_bytes(pt[48:], "little")
    return pt[:48] + ((FQ - y) % FQ).to_bytes(48, "little")


G = z.g1_generator()
P2, P3 = z.g1_mul(G, (2).to_bytes(32, "little")), z.g1_mul(G, (3).to_bytes(32, "little"))
pts = [G, G, g1_neg(G), G, None, P2, g1_neg(P2), P3, P3]
kk = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890AB % FR
for reps, scal in ((1, [kk] * 9), (1, [kk, kk, kk, FR - 1, 5, 0, 1, FR - 1, FR - 1]), (400, [kk] * 9), (37, [kk, 1, kk, 1, 0, 1, 1, kk, FR - 1])):
    raw_b = b"".join(pt if pt is not None else bytes(96) for pt in pts * reps)
    raw_s = b"".join(int(v).to_bytes(32, "little") for v in scal * reps)
    for prepared in (False, True):
        bb = ctx.bases_g1(raw_b)
        if prepared:
            bb.prepare()
        h.update(ctx.msm_g1(raw_s, bb))
        bb.free()
for lg, count in ((13, 70), (17, 9)):
    r1, wits = bench.relation_and_witness(z, "poseidon", lg, [lg, lg + 1])
    prng = bench.SplitMix64(lg)
    toxic = b"".join(prng.fr_bytes() for _ in range(5))
    pk, vk = ctx.groth16_setup(r1, toxic)
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
    if lg == 13:
        # the same relation with one-proof groups (what a 2^20 key runs: L, H and B1 -- taken over r z -- share one
        # reduction): seven proofs of a pipelined batch = the first seven of the grouped batch
        ctx.set_group_size(1)
        pk1, vk1 = ctx.groth16_setup(r1, toxic)
        ctx.set_group_size(0)
        assert vk1 == vk
        one = ctx.groth16_prove_batch_dev(pk1, [d[i % 2].data_ptr() for i in range(7)], rs[:7], ss[:7])
        assert one == proofs[:7]
        for p in one:
            h.update(p)
        pk1.free()
    r1.free()
ctx.close()
print("VARIANT_DIGEST", h.hexdigest())
