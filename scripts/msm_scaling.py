"""zkmi_msm_g1_dev for n = 2^lo .. 2^hi (uniform scalars, synthetic bases), wall clock and per-phase HIP-event times.
Usage: python scripts/msm_scaling.py [lo [hi [prepared [witness]]]]  (defaults 20 26 0 0; prepared = 1: zkmi_bases_g1_prepare first;
witness = 1: the witness-like mix -- 40 % zeros, 20 % ones, 40 % full-width -- instead of uniform scalars)"""
import os
import sys
import time

sys.path.insert(0, ".")
import torch

from zkmi_loader import load_pkg

pkg = load_pkg()
z = pkg.Zkmi(os.environ.get("ZKMI_LIB"))
ctx = z.context(0)
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 26
prepared = len(sys.argv) > 3 and sys.argv[3] == "1"
witness = len(sys.argv) > 4 and sys.argv[4] == "1"
print(f"{'log_n':>5} {'wall ms':>9} {'G terms/s':>9}  phases (ms)")
for lg in range(lo, hi + 1):
    n = 1 << lg
    g = torch.Generator(device="cuda").manual_seed(lg)
    raw = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    raw[:, 31] &= 0x3F
    if witness:
        kind = torch.rand(n, device="cuda", generator=g)
        raw[kind < 0.6] = 0
        raw[(kind >= 0.4) & (kind < 0.6), 0] = 1
        del kind
    b = ctx.bases_g1_synthetic(n)
    if prepared:
        t0 = time.perf_counter()
        b.prepare()
        print(f"      table for 2^{lg} bases built in {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
    torch.cuda.synchronize()
    ctx.msm_g1_dev(raw.data_ptr(), n, b)
    ctx.prof_enable(True)
    ctx.prof_reset()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.msm_g1_dev(raw.data_ptr(), n, b)
        ts.append(time.perf_counter() - t0)
    ctx.prof_enable(False)
    ph = {k: round(ctx.prof_get(k)[0] / 3, 2) for k in ("msm_sort", "msm_accum_g1", "msm_reduce_g1")}
    t = sorted(ts)[1]
    print(f"{lg:>5} {1e3 * t:>9.2f} {n / t / 1e9:>9.3f}  {ph}", flush=True)
    b.free()
    del raw
    torch.cuda.empty_cache()
