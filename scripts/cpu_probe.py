"""What the host gives this container: logical CPUs, affinity mask, cgroup CPU quota; and how the oracle's G1 MSM
(2^18 terms) scales with the thread count asked for.  Usage: python scripts/cpu_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.strerror)
import json
import random

from oracle import cpp as ocpp

ocpp.build()
print("oracle_threads", ocpp.threads())
g = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "msm_small.json")))
v = [x for x in g if x["group"] == 1 and len(x["bases"]) // 192 >= 16][0]
bases = bytes.fromhex(v["bases"])
nb = len(bases) // 96
n = 1 << 18
rnd = random.Random(5)
B = b"".join(bases[96 * (i % nb) : 96 * (i % nb) + 96] for i in range(n))
S = bytearray(rnd.randbytes(32 * n))
for i in range(31, 32 * n, 32):
    S[i] &= 0x3F
S = bytes(S)
for t in (1, 4, 8, 16, 17, 32, 64, 128, 256):
    t0 = time.time()
    ocpp.msm_g1(S, B, t)
    print("threads", t, "seconds %.3f" % (time.time() - t0), flush=True)
