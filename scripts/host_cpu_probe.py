#!/usr/bin/env python3
"""Where the host's CPU time goes while the batch prover runs: per-thread user + system time (/proc/self/task/*/stat) around a
batch of proofs, at 2^20 (pipelined one-proof groups) and at 2^14 (groups of 64).
    python scripts/host_cpu_probe.py [log_n ...]          (ZKMI_LIB=zk-apps_amd/libzkmi_exp.so ZKMI_HOST_WAIT=0: spinning waits)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

TICK = os.sysconf("SC_CLK_TCK")


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/stat") as f:
                raw = f.read()
            name = raw[raw.index("(") + 1: raw.rindex(")")]
            f_ = raw[raw.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(f_[11]) + int(f_[12])) / TICK)
        except (OSError, ValueError):
            pass
    return out


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [20, 14]
    z = bench.load_pkg().Zkmi()
    ctx = z.context(0)
    print("host:", z.host_info(), "lib:", os.environ.get("ZKMI_LIB", "product"), "ZKMI_HOST_WAIT=%s" % os.environ.get("ZKMI_HOST_WAIT", "default"))
    for lg in sizes:
        count = 40 if lg >= 19 else 1024 if lg <= 15 else 256
        r1, wits = bench.relation_and_witness(z, "poseidon", lg, [7, 8])
        rng = bench.SplitMix64(lg)
        pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
        d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
        rs = [rng.fr_bytes() for _ in range(count)]
        ss = [rng.fr_bytes() for _ in range(count)]
        ptrs = [d[i % 2].data_ptr() for i in range(count)]
        torch.cuda.synchronize()
        ctx.groth16_prove_batch_dev(pk, ptrs[: max(4, count // 8)], rs[: max(4, count // 8)], ss[: max(4, count // 8)])
        ctx.sync()
        t_before = threads()
        c0, t0 = bench.cpu_seconds(), time.perf_counter()
        proofs = ctx.groth16_prove_batch_dev(pk, ptrs, rs, ss)
        ctx.sync()
        dt, cpu = time.perf_counter() - t0, bench.cpu_seconds() - c0
        t_after = threads()
        assert z.groth16_verify(vk, wits[(count - 1) % 2][32: 32 * r1.n_pub], proofs[-1])
        print(f"2^{lg}: {count} proofs in {dt:.3f} s = {count / dt:.1f} proofs/s; CPU {cpu:.3f} s = {1e3 * cpu / count:.3f} ms per proof, "
              f"{cpu / dt:.2f} CPUs busy (x8 ranks: {8 * cpu / dt:.1f})")
        rows = []
        for tid, (name, v) in t_after.items():
            dv = v - t_before.get(tid, (name, 0.0))[1]
            if dv > 0.0:
                rows.append((dv, tid, name))
        for dv, tid, name in sorted(rows, reverse=True)[:12]:
            print(f"    thread {tid:>8} {name:<18} {dv:7.3f} s  = {dv / dt:5.2f} CPU")
        pk.free()
        r1.free()
    ctx.close()


if __name__ == "__main__":
    main()
