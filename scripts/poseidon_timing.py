"""Ad-hoc timing of the Poseidon-5 kernels (2-to-1 hashes and a Merkle tree), HIP-event timed."""
import sys
import time

sys.path.insert(0, ".")
import torch

from zkmi_loader import load_pkg

pkg = load_pkg()
z = pkg.Zkmi()
ctx = z.context(0)
ctx.prof_enable(True)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(0, 256, (n, 2, 32), dtype=torch.uint8, device="cuda", generator=g)
raw[:, :, 31] &= 0x3F
out = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for field in (0, 1):
    for it in range(3):
        ctx.prof_reset()
        t = time.time(); ctx.poseidon_hash_batch_dev(raw.data_ptr(), n, 2, out.data_ptr(), field); dt = time.time() - t
        ms = ctx.prof_get("witness")[0]
        print(f"field {field}: 2^{lg} two-to-one hashes: wall {dt*1e3:.2f} ms, kernel {ms:.3f} ms, {n/ms/1e3:.1f} M hashes/s")
nodes = torch.zeros((2 * n - 1, 32), dtype=torch.uint8, device="cuda")
nodes[:n] = out
torch.cuda.synchronize()
for it in range(2):
    ctx.prof_reset()
    t = time.time(); ctx.poseidon_merkle_tree_dev(nodes.data_ptr(), lg); dt = time.time() - t
    print(f"merkle tree 2^{lg} leaves: wall {dt*1e3:.2f} ms, kernels {ctx.prof_get('witness')[0]:.3f} ms")
