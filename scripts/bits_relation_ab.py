#!/usr/bin/env python3
"""What the r B1 fold costs a witness of BITS: a relation of b_i * b_i = b_i rows, 60 % ones -- the digits of z
fill one digit position, the digits of r z all thirteen, and every 1 becomes the same scalar r (one heavy bucket per
position).  Proof rate of a pipelined batch through zkmi_groth16_prove_batch_dev; run once per setting of the A/B library:
    ZKMI_LIB=zk-apps_amd/libzkmi_exp.so ZKMI_RB1_FOLD=0|2 python scripts/bits_relation_ab.py [log_n = 18] [proofs = 24] [sum]
("sum" adds one row that sums all the bits: a linear combination of 2^log_n terms.)"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    z = bench.load_pkg().Zkmi(os.environ.get("ZKMI_LIB"))
    ctx = z.context(0)
    one = (1).to_bytes(32, "little")
    n_pub = 2
    nbits = (1 << lg) - n_pub - 8
    n_vars = n_pub + nbits
    rnd = random.Random(lg)
    bits = [1 if rnd.random() < 0.6 else 0 for _ in range(nbits)]
    zv = [1, bits[0]] + bits
    cols = list(range(n_pub, n_pub + nbits))
    rp = list(range(nbits + 2))
    a = (rp, cols + [n_pub], one * (nbits + 1))  # rows b_i * b_i = b_i, last row b_0 * 1 = z[1] (the public input)
    b = (rp, cols + [0], one * (nbits + 1))
    c = (rp, cols + [1], one * (nbits + 1))
    if len(sys.argv) > 3 and sys.argv[3] == "sum":
        # the public input becomes the NUMBER of ones: its row sums all the bits, (b_0 + b_1 + ...) * 1 = z[1] -- a linear
        # combination of nbits terms, a row one lane cannot take (k_matvec_long's)
        zv[1] = sum(bits)
        a = (rp[:-1] + [2 * nbits], cols + cols, one * (2 * nbits))
    r1 = z.r1cs_create(n_vars, n_pub, [a, b, c])
    wit = b"".join(v.to_bytes(32, "little") for v in zv)
    assert r1.is_satisfied(wit), "relation"
    prng = bench.SplitMix64(lg)
    pk, vk = ctx.groth16_setup(r1, b"".join(prng.fr_bytes() for _ in range(5)))
    d = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
    rs = [prng.fr_bytes() for _ in range(count)]
    ss = [prng.fr_bytes() for _ in range(count)]
    torch.cuda.synchronize()
    ctx.groth16_prove_batch_dev(pk, [d.data_ptr()] * 4, rs[:4], ss[:4])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        proofs = ctx.groth16_prove_batch_dev(pk, [d.data_ptr()] * count, rs, ss)
        best = min(best, time.perf_counter() - t0)
    ok = z.groth16_verify(vk, wit[32: 32 * n_pub], proofs[-1])
    t0 = time.perf_counter()
    ctx.groth16_prove_dev(pk, d.data_ptr(), rs[0], ss[0])
    one_ms = (time.perf_counter() - t0) * 1e3
    print(f"bits relation 2^{r1.log_n} (n_vars {n_vars}, {sum(bits)} ones) fold={os.environ.get('ZKMI_RB1_FOLD', 'default')}: "
          f"{count / best:8.1f} proofs/s  {best / count * 1e3:7.3f} ms/proof  one proof alone {one_ms:.2f} ms  verified {ok}", flush=True)


if __name__ == "__main__":
    main()
