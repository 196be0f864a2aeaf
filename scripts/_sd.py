import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
z = bench.load_pkg().Zkmi()
ctx = z.context(0)
lg = int(sys.argv[1])
r = bench.small_domain_rate(z, ctx, "poseidon", lg, 1024 if lg <= 14 else 256)
print(r)
