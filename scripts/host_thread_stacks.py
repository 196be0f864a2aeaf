#!/usr/bin/env python3
"""Which thread of a proving process burns a CPU: runs a pipelined batch in a background thread, samples per-thread CPU, and
asks rocgdb (attached to this very process: PR_SET_PTRACER_ANY) for the backtrace of the busiest non-main threads."""
import ctypes
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402
from host_cpu_probe import threads  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    z = bench.load_pkg().Zkmi()
    ctx = z.context(0)
    r1, wits = bench.relation_and_witness(z, "poseidon", lg, [7, 8])
    rng = bench.SplitMix64(lg)
    pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
    d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
    count = 3000
    rs = [rng.fr_bytes() for _ in range(count)]
    ss = [rng.fr_bytes() for _ in range(count)]
    ptrs = [d[i % 2].data_ptr() for i in range(count)]
    torch.cuda.synchronize()
    ctx.groth16_prove_batch_dev(pk, ptrs[:8], rs[:8], ss[:8])
    done = []
    th = threading.Thread(target=lambda: done.append(ctx.groth16_prove_batch_dev(pk, ptrs, rs, ss)))
    idle0 = threads()
    time.sleep(1.0)
    idle1 = threads()
    print("idle second (nothing queued):", sorted(((v[1] - idle0.get(t, v)[1], t) for t, v in idle1.items()), reverse=True)[:4])
    th.start()
    time.sleep(1.0)
    a = threads()
    time.sleep(1.0)
    b = threads()
    busy = sorted(((v[1] - a.get(t, v)[1], t) for t, v in b.items()), reverse=True)[:4]
    print("busiest threads over one second of proving:", busy, "main tid", os.getpid())
    ctypes.CDLL(None).prctl(0x59616D61, ctypes.c_ulong(-1 & 0xFFFFFFFFFFFFFFFF), 0, 0, 0)  # PR_SET_PTRACER, PR_SET_PTRACER_ANY
    cmds = []
    for _, tid in busy[:3]:
        cmds += ["-ex", "thread find %d" % tid]
    out = subprocess.run(["/opt/rocm/bin/rocgdb", "-p", str(os.getpid()), "-batch", "-ex", "set pagination off", "-ex", "info threads",
                          "-ex", "thread apply all bt 12"], capture_output=True, text=True, timeout=300)
    txt = out.stdout + out.stderr
    # keep the blocks of the busy threads
    for _, tid in busy[:3]:
        key = "LWP %d" % tid
        for blk in txt.split("\nThread "):
            if key in blk.split("\n")[0]:
                print("---- Thread " + "\n".join(blk.split("\n")[:16]))
    if "LWP" not in txt:
        print(txt[-3000:])
    th.join()
    ctx.close()


if __name__ == "__main__":
    main()
