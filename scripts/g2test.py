import os, sys
sys.path.insert(0, ".")
from zkmi_loader import load_pkg
from oracle import bls12_381 as ec
from oracle.bls12_381 import R
pkg = load_pkg()
z = pkg.Zkmi(os.environ.get("ZKMI_LIB"))
ctx = z.context(0)
for n in (300, 2100, 5000, 16000, 20000):
    rng = ec.SplitMix64(7 * n)
    s = [rng.fr() for _ in range(n)]
    b = ctx.bases_g2_synthetic(n)
    got = ctx.msm_g2(b"".join(ec.fr_to_bytes(v) for v in s), b)
    q = ec.g2_mul(0xC0FFEE)
    exp = ec.pt_add(ec.Fq2, ec.g2_mul(sum(s) % R), ec.g2_mul(sum(i * v for i, v in enumerate(s)) % R, q))
    print(n, got == ec.g2_to_bytes(exp))
