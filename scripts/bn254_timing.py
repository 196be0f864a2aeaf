"""Ad-hoc timing of the BN254 MSM / NTT kernels (KZG-commit-shaped driver), HIP-event timed."""
import sys
import time

sys.path.insert(0, ".")
import torch

from zkmi_loader import load_pkg

pkg = load_pkg()
z = pkg.Zkmi(__import__("os").environ.get("ZKMI_LIB"))
ctx = z.context(0)
ctx.prof_enable(True)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
raw[:, 31] &= 0x1F
torch.cuda.synchronize()
t = time.time(); b = ctx.bn254_bases_synthetic(n); print("synthetic bases s", round(time.time() - t, 2))
for it in range(3):
    ctx.prof_reset()
    t = time.time(); ctx.bn254_msm_g1_dev(raw.data_ptr(), n, b); dt = time.time() - t
    print(f"bn254 msm_g1 2^{lg}: wall {dt*1e3:.2f} ms", {k: round(ctx.prof_get(k)[0], 3) for k in ("msm_sort", "msm_accum_g1", "msm_reduce_g1")})
b.prepare()
for it in range(3):
    ctx.prof_reset()
    t = time.time(); ctx.bn254_msm_g1_dev(raw.data_ptr(), n, b); dt = time.time() - t
    print(f"bn254 msm_g1 2^{lg} (prepared SRS): wall {dt*1e3:.2f} ms", {k: round(ctx.prof_get(k)[0], 3) for k in ("msm_sort", "msm_accum_g1", "msm_reduce_g1")})
x = raw.clone()
for it in range(3):
    ctx.prof_reset()
    t = time.time(); ctx.bn254_ntt_dev(x.data_ptr(), lg); dt = time.time() - t
    print(f"bn254 ntt 2^{lg}: wall {dt*1e3:.3f} ms, transform {ctx.prof_get('ntt')[0]:.3f} ms")
for it in range(2):
    ctx.prof_reset()
    t = time.time(); ctx.bn254_kzg_commit_dev(x.data_ptr(), lg, b); dt = time.time() - t
    print(f"bn254 kzg commit from 2^{lg} evaluations: wall {dt*1e3:.2f} ms")
zeta = (0x2B1C5E9F00D1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234 % 21888242871839275222246405745257275088548364400416034343698204186575808495617).to_bytes(32, "little")
for it in range(3):
    ctx.prof_reset()
    t = time.time(); ctx.bn254_kzg_open_dev(raw.data_ptr(), n, zeta, b); dt = time.time() - t
    print(f"bn254 kzg open of 2^{lg} coefficients (prepared SRS): wall {dt*1e3:.2f} ms, evaluation + quotient scan {ctx.prof_get('misc')[0]:.3f} ms,",
          {k: round(ctx.prof_get(k)[0], 3) for k in ("msm_sort", "msm_accum_g1", "msm_reduce_g1")})
out = torch.empty_like(raw)
den = raw.clone(); den[:, 0] |= 1
torch.cuda.synchronize()
for it in range(3):
    ctx.prof_reset()
    t = time.time(); ctx.bn254_grand_product_dev(raw.data_ptr(), den.data_ptr(), n, out.data_ptr()); dt = time.time() - t
    print(f"bn254 grand product of 2^{lg} terms: wall {dt*1e3:.2f} ms, kernels {ctx.prof_get('misc')[0]:.3f} ms")
polys = [raw] + [torch.roll(raw, shifts=7 * (j + 1), dims=0) for j in range(7)]
torch.cuda.synchronize()
for it in range(3):
    t = time.time(); ctx.bn254_kzg_open_many_dev([p.data_ptr() for p in polys], n, zeta, zeta, b); dt = time.time() - t
    print(f"bn254 kzg open of 8 polynomials of 2^{lg} coefficients at one point (prepared SRS): wall {dt*1e3:.2f} ms")
