"""A few single proofs (zkmi_groth16_prove_dev, one at a time) at N = 2^lg: the target of
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/single_proof_trace.py LG
followed by scripts/trace_timeline.py on the last few ms (the latency chain of one small proof)."""
import sys
import time

sys.path.insert(0, ".")
import torch

import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
import os
z = bench.load_pkg().Zkmi(os.environ.get("ZKMI_LIB"))
ctx = z.context(0)
r1, wits = bench.relation_and_witness(z, "poseidon", lg, [1, 2])
rng = bench.SplitMix64(3)
pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
r, s = rng.fr_bytes(), rng.fr_bytes()
torch.cuda.synchronize()
for i in range(4):
    ctx.groth16_prove_dev(pk, d[i % 2].data_ptr(), r, s)
ctx.sync()
time.sleep(0.05)
lat = []
for i in range(3):
    t0 = time.perf_counter()
    p = ctx.groth16_prove_dev(pk, d[i % 2].data_ptr(), r, s)
    lat.append(1e3 * (time.perf_counter() - t0))
    time.sleep(0.02)
print("latencies ms", [round(x, 2) for x in lat], "verified", z.groth16_verify(vk, wits[0][32 : 32 * r1.n_pub], p), flush=True)
