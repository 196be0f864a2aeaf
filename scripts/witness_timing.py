"""Ad-hoc timing of device-side update_note assignment generation (one GPU thread per instance)."""
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from zkmi_loader import load_pkg

pkg = load_pkg()
z = pkg.Zkmi()
ctx = z.context(0)
ctx.prof_enable(True)


def make(seed):
    rng = bench.SplitMix64(seed)
    f = lambda: int.from_bytes(rng.fr_bytes(), "little")
    tok = (f(), f())
    bal = (rng.next() >> 1, rng.next() >> 1)
    user = f()
    return z.note_update(bal[0] >> 3, tok[0], user, (f(), f(), f()), (f(), f(), f()), [rng.next() & 1 for _ in range(10)],
                         [f() for _ in range(10)], user, (tok[0], bal[0], tok[1], bal[1]))


for lg, batches in ((14, (64, 1024, 8192)), (20, (1, 64, 512))):
    n = 1 << lg
    for B in batches:
        ins = [make(1000 + i) for i in range(B)]
        big = torch.empty((B, 32 * n), dtype=torch.uint8, device="cuda")
        ptrs = [big[i].data_ptr() for i in range(B)]
        torch.cuda.synchronize()
        for it in range(2):
            ctx.prof_reset()
            t = time.time(); st = ctx.update_note_witness_batch_dev(lg, 1, ins, ptrs); dt = time.time() - t
            ms = ctx.prof_get("witness")[0]
        assert st == [0] * B
        host = z.update_note_witness(lg, 1, ins[-1])[0]
        assert bytes(big[-1].cpu().numpy().tobytes()) == host
        print(f"N=2^{lg} batch {B}: kernel {ms:.1f} ms, wall {dt*1e3:.1f} ms -> {B/ms*1e3:.0f} assignments/s")
        del big
