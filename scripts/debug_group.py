import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zkmi_loader import load_pkg
from oracle import bls12_381 as ec
from test_cpu_host import _note_update_case
lg = int(sys.argv[1]); count = int(sys.argv[2])
pkg = load_pkg(); zk = pkg.Zkmi(); ctx = zk.context(0)
frs = lambda vals: b"".join(ec.fr_to_bytes(v) for v in vals)
r1 = zk.update_note_r1cs(lg, 1)
rng = ec.SplitMix64(77)
toxic = frs([rng.fr() for _ in range(5)])
cases = [_note_update_case(zk, 9000 + i, 1, amount=1 + i % 7, balances=(100 + i, 9)) for i in range(count)]
bufs = [torch.zeros(32 << lg, dtype=torch.uint8, device="cuda") for _ in cases]
torch.cuda.synchronize()
for k in range(0, count, 64):
    ch = list(range(k, min(count, k + 64)))
    assert ctx.update_note_witness_batch_dev(lg, 1, [cases[i][0] for i in ch], [bufs[i].data_ptr() for i in ch]) == [0] * len(ch)
rs = [ec.fr_to_bytes(rng.fr()) for _ in range(count)]; ss = [ec.fr_to_bytes(rng.fr()) for _ in range(count)]
pk, vk = ctx.groth16_setup(r1, toxic)
for rep in range(3):
    grouped = ctx.groth16_prove_batch_dev(pk, [b.data_ptr() for b in bufs], rs, ss)
    bad = [i for i, ((_, pub), pf) in enumerate(zip(cases, grouped)) if not zk.groth16_verify(vk, frs(pub), pf)]
    print("rep", rep, "bad proofs:", bad)
ctx.set_group_size(1)
pk1, vk1 = ctx.groth16_setup(r1, toxic)
single = ctx.groth16_prove_batch_dev(pk1, [b.data_ptr() for b in bufs], rs, ss)
print("differs from one-by-one:", [i for i in range(count) if grouped[i] != single[i]])
