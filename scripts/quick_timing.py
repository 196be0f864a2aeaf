"""Ad-hoc timing of the raw kernels (not the driver's bench contract; see bench.py)."""
import random
import sys
import time

sys.path.insert(0, ".")
import torch

from zkmi_loader import load_pkg

pkg = load_pkg()
import os
z = pkg.Zkmi(os.environ.get("ZKMI_LIB"))
ctx = z.context(0)
ctx.prof_enable(True)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
rnd = random.Random(1)
raw = bytearray(rnd.randbytes(32 * n))
for i in range(31, 32 * n, 32):
    raw[i] &= 0x3F
d = torch.frombuffer(raw, dtype=torch.uint8).cuda()
torch.cuda.synchronize()
t = time.time(); b1 = ctx.bases_g1_synthetic(n); print("synthetic g1 bases s", time.time() - t)
for it in range(3):
    ctx.prof_reset()
    t = time.time(); out = ctx.msm_g1_dev(d.data_ptr(), n, b1); dt = time.time() - t
    print(f"msm_g1 2^{lg}: wall {dt*1e3:.2f} ms", {k: ctx.prof_get(k) for k in ("msm_sort", "msm_accum_g1", "msm_reduce_g1")})
if lg <= 20:
    t = time.time(); b2 = ctx.bases_g2_synthetic(n); print("synthetic g2 bases s", time.time() - t)
    for it in range(2):
        ctx.prof_reset()
        t = time.time(); out = ctx.msm_g2_dev(d.data_ptr(), n, b2); dt = time.time() - t
        print(f"msm_g2 2^{lg}: wall {dt*1e3:.2f} ms", {k: ctx.prof_get(k) for k in ("msm_sort", "msm_accum_g2", "msm_reduce_g2")})
x = d.clone()
for it in range(3):
    ctx.prof_reset()
    t = time.time(); ctx.ntt_dev(x.data_ptr(), lg); dt = time.time() - t
    print(f"ntt 2^{lg}: wall {dt*1e3:.3f} ms", ctx.prof_get("ntt"))
# witness-like scalars (SURVEY §8d secondary): 40 % zero, 20 % one, 10 % < 2^16, 30 % uniform
import numpy as np
rs = np.random.RandomState(3)
arr = np.frombuffer(bytes(raw), dtype=np.uint8).reshape(n, 32).copy()
kind = rs.randint(0, 10, size=n)
arr[kind < 4] = 0
ones = (kind >= 4) & (kind < 6)
arr[ones] = 0
arr[ones, 0] = 1
small = kind == 6
arr[small, 2:] = 0
dw = torch.from_numpy(arr).cuda()
torch.cuda.synchronize()
for it in range(3):
    ctx.prof_reset()
    t = time.time(); out = ctx.msm_g1_dev(dw.data_ptr(), n, b1); dt = time.time() - t
    print(f"msm_g1 witness-like 2^{lg}: wall {dt*1e3:.2f} ms", {k: ctx.prof_get(k) for k in ("msm_sort", "msm_accum_g1", "msm_reduce_g1")})
