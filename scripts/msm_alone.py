"""One G1 MSM of 2^20 terms, alone on the chip, a few times: the target of
   rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU -- python3 scripts/msm_alone.py
(kernel durations and counters in the same run: resident waves per SIMD and the clock the chip holds)."""
import os
import sys

sys.path.insert(0, ".")
import torch

from zkmi_loader import load_pkg

pkg = load_pkg()
z = pkg.Zkmi(os.environ.get("ZKMI_LIB"))
ctx = z.context(0)
n = 1 << 20
g = torch.Generator(device="cuda").manual_seed(5)
raw = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
raw[:, 31] &= 0x3F
b = ctx.bases_g1_synthetic(n)
b2 = ctx.bases_g2_synthetic(n)
torch.cuda.synchronize()
for _ in range(4):
    ctx.msm_g1_dev(raw.data_ptr(), n, b)
for _ in range(2):
    ctx.msm_g2_dev(raw.data_ptr(), n, b2)
