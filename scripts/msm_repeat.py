"""The same windowed G1 MSM again and again: every result must be the same 96 bytes.  (The two-level digit sort of msm_sort.hip
reserves a batch's room in a fine partition with a global atomic, so the ORDER of a bucket's entries differs from run to run;
the sums must not.)  Usage: python scripts/msm_repeat.py [log_n [repeats [witness]]]  (defaults 24 12 0)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from zkmi_loader import load_pkg  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
witness = len(sys.argv) > 3 and sys.argv[3] == "1"
z = load_pkg().Zkmi(os.environ.get("ZKMI_LIB"))
ctx = z.context(0)
n = (1 << lg) - 3
g = torch.Generator(device="cuda").manual_seed(lg)
raw = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
raw[:, 31] &= 0x3F
if witness:
    kind = torch.rand(n, device="cuda", generator=g)
    raw[kind < 0.6] = 0
    raw[(kind >= 0.4) & (kind < 0.6), 0] = 1
    del kind
b = ctx.bases_g1_synthetic(n)
torch.cuda.synchronize()
seen = {}
for _ in range(reps):
    r = ctx.msm_g1_dev(raw.data_ptr(), n, b)
    seen[hashlib.sha256(r).hexdigest()[:16]] = seen.get(hashlib.sha256(r).hexdigest()[:16], 0) + 1
print(f"2^{lg} - 3 terms ({'witness-like' if witness else 'uniform'}), {reps} runs: {len(seen)} distinct result(s) {seen}", flush=True)
sys.exit(0 if len(seen) == 1 else 1)
