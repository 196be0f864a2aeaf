import sys
sys.path.insert(0, ".")
import torch
import bench
z = bench.load_pkg().Zkmi()
ctx = z.context(0)
free0, tot = torch.cuda.mem_get_info()
for lg in (14, 20):
    r1, wits = bench.relation_and_witness(z, "poseidon", lg, [1])
    rng = bench.SplitMix64(1)
    f_before = torch.cuda.mem_get_info()[0]
    pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
    d = torch.frombuffer(bytearray(wits[0]), dtype=torch.uint8).cuda()
    ctx.groth16_prove_batch_dev(pk, [d.data_ptr()] * 4, [rng.fr_bytes()] * 4, [rng.fr_bytes()] * 4)
    f_after = torch.cuda.mem_get_info()[0]
    print(f"2^{lg}: key + workspaces after first proofs: {(f_before - f_after) / 2**30:.2f} GiB (device total {tot / 2**30:.0f} GiB)")
    pk.free(); r1.free()
    print(f"   after pk.free: still held by the context (workspaces): {(free0 - torch.cuda.mem_get_info()[0]) / 2**30:.2f} GiB")
