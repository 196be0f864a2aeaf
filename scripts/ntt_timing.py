"""Fr NTT timing on an idle GPU: HIP-event time of the transform's passes (zkmi_prof, phase "ntt") for the public
entry point, per size.  ZKMI_LIB=zk-apps_amd/libzkmi_exp.so ZKMI_NTT_RB=0|1|2 selects the pass kernels of the A/B library (see ntt.hip).  Usage: python scripts/ntt_timing.py [lg ...]"""
import sys

sys.path.insert(0, ".")
import torch

from zkmi_loader import load_pkg

pkg = load_pkg()
z = pkg.Zkmi()
ctx = z.context(0)
ctx.prof_enable(True)
for lg in [int(a) for a in sys.argv[1:]] or [14, 16, 18, 20, 22]:
    n = 1 << lg
    x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    x[:, 31] &= 0x3F
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.ntt_dev(x.data_ptr(), lg)
    ctx.prof_reset()
    reps = 20
    for _ in range(reps):
        ctx.ntt_dev(x.data_ptr(), lg)
    ms, cnt = ctx.prof_get("ntt")
    print(f"ntt 2^{lg}: {ms / cnt:.4f} ms per transform (bit-reversal copy + passes), {64.0 * n / (ms / cnt * 1e-3) / 1e9:.0f} GB/s algorithmic", flush=True)
ctx.close()
