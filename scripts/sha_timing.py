"""Ad-hoc timing of the SHA-256 pair-hash kernel and the contract-semantics Merkle tree."""
import sys
import time

sys.path.insert(0, ".")
import torch

from zkmi_loader import load_pkg

pkg = load_pkg()
z = pkg.Zkmi()
ctx = z.context(0)
ctx.prof_enable(True)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for it in range(4):
    ctx.prof_reset()
    ctx.sha256_pairs_dev(raw.data_ptr(), n, out.data_ptr())
    ms = ctx.prof_get("misc")[0]
    print(f"2^{lg} pair hashes: {ms:.3f} ms, {n/ms/1e3:.0f} M hashes/s, {96*n/ms/1e6:.0f} GB/s algorithmic")
nodes = torch.zeros((2 * n - 1, 32), dtype=torch.uint8, device="cuda")
nodes[:n] = out
torch.cuda.synchronize()
for it in range(2):
    ctx.prof_reset()
    ctx.sha256_merkle_tree_dev(nodes.data_ptr(), lg, n)
    print(f"full tree over 2^{lg} leaves: {ctx.prof_get('misc')[0]:.3f} ms")
