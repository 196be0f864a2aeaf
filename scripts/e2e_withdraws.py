"""End to end on one GPU: note tree (Poseidon) -> Merkle paths -> update_note assignments generated on the
device -> pipelined batch of Groth16 proofs -> pairing verification of a sample.  No oracle involved.
Usage: python scripts/e2e_withdraws.py [log_n=16] [batch=64]"""
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from zkmi_loader import load_pkg

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pkg = load_pkg()
z = pkg.Zkmi()
ctx = z.context(0)
rng = bench.SplitMix64(2026)
f = lambda: int.from_bytes(rng.fr_bytes(), "little")
fb = lambda v: int(v).to_bytes(32, "little")

t0 = time.time()
r1 = z.update_note_r1cs(lg, 1)
pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
print(f"relation + trusted setup at N=2^{lg}: {time.time() - t0:.2f} s (one time)")

# B accounts with one old note each, inserted in a 1024-leaf Poseidon note tree built on the device
DEPTH = 10
accounts = [dict(tok=(f(), f()), bal=(rng.next() >> 2, rng.next() >> 2), old=(f(), f(), f()), new=(f(), f(), f()), user=f())
            for _ in range(B)]
t0 = time.time()
acc_vec = b"".join(fb(a["tok"][0]) + fb(a["bal"][0]) + fb(a["tok"][1]) + fb(a["bal"][1]) for a in accounts)
acc_hash = ctx.poseidon_hash_batch(acc_vec, B, 4)
note_vec = b"".join(fb(a["old"][0]) + fb(a["old"][1]) + fb(a["old"][2]) + acc_hash[32 * i : 32 * i + 32] for i, a in enumerate(accounts))
note_hash = ctx.poseidon_hash_batch(note_vec, B, 4)
n_leaves = 1 << DEPTH
nodes = torch.zeros((2 * n_leaves - 1, 32), dtype=torch.uint8, device="cuda")
slots = [(7 * i + 3) % n_leaves for i in range(B)]
assert len(set(slots)) == B or B > n_leaves  # more withdraws than leaves: notes share leaves (timing runs)
leaf_buf = bytearray(32 * n_leaves)
for i, s in enumerate(slots):
    leaf_buf[32 * s : 32 * s + 32] = note_hash[32 * i : 32 * i + 32]
nodes[:n_leaves] = torch.frombuffer(leaf_buf, dtype=torch.uint8).view(n_leaves, 32).cuda()
torch.cuda.synchronize()
ctx.poseidon_merkle_tree_dev(nodes.data_ptr(), DEPTH)
shape, paths = ctx.poseidon_merkle_paths_dev(nodes.data_ptr(), DEPTH, slots)
root = int.from_bytes(bytes(nodes[-1].cpu().numpy().tobytes()), "little")
t_tree = time.time() - t0

ints = lambda raw: [int.from_bytes(raw[k : k + 32], "little") for k in range(0, len(raw), 32)]
inputs = []
for i, a in enumerate(accounts):
    amount = a["bal"][0] >> 4
    inputs.append(z.note_update(amount, a["tok"][0], a["user"], a["new"], a["old"], list(shape[DEPTH * i : DEPTH * (i + 1)]),
                                ints(paths[32 * DEPTH * i : 32 * DEPTH * (i + 1)]), a["user"],
                                (a["tok"][0], a["bal"][0], a["tok"][1], a["bal"][1])))
n = 1 << lg
bufs = torch.empty((B, 32 * n), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.time()
status = ctx.update_note_witness_batch_dev(lg, 1, inputs, [bufs[i].data_ptr() for i in range(B)])
t_wit = time.time() - t0
assert status == [0] * B
rs = [rng.fr_bytes() for _ in range(B)]
ss = [rng.fr_bytes() for _ in range(B)]
t0 = time.time()
proofs = ctx.groth16_prove_batch_dev(pk, [bufs[i].data_ptr() for i in range(B)], rs, ss)
t_prove = time.time() - t0
t0 = time.time()
ok = 0
for i in (0, B // 2, B - 1):
    pub = bytes(bufs[i][32 : 32 * 7].cpu().numpy().tobytes())
    assert int.from_bytes(pub[128:160], "little") == root  # public merkle_root = root of the device-built tree
    ok += z.groth16_verify(vk, pub, proofs[i])
t_ver = (time.time() - t0) / 3
print(f"batch of {B} withdraws at N=2^{lg}: tree+paths {t_tree*1e3:.1f} ms, assignments on device {t_wit*1e3:.1f} ms, "
      f"proofs {t_prove*1e3:.1f} ms ({B/t_prove:.1f} proofs/s), verify {t_ver*1e3:.1f} ms each; "
      f"end to end {B/(t_tree+t_wit+t_prove):.1f} withdraws/s; sample verified: {ok}/3")
