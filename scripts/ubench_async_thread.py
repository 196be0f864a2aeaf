#!/usr/bin/env python3
"""What keeps ROCr's AsyncEventsLoop thread busy (it runs at 1.00 CPU beside a proving process): tiny kernels on one or two
streams with and without event records / cross-stream waits, per-thread CPU from /proc.  PyTorch's bundled runtime (what
bench.py runs under)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from host_cpu_probe import threads  # noqa: E402


def run(name, n, body):
    torch.cuda.synchronize()
    a, t0 = threads(), time.perf_counter()
    body(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b = threads()
    me = os.getpid()
    rows = sorted(((v[1] - a.get(t, v)[1], t) for t, v in b.items()), reverse=True)
    main = next((d for d, t in rows if t == me), 0.0)
    other = [(round(d, 2), t) for d, t in rows if t != me and d > 0.0][:3]
    print(f"{name:<58} {n / dt / 1e3:7.1f} k iters/s  wall {dt:5.2f} s  main {main:5.2f} s  others {other}", flush=True)


def main():
    x = torch.zeros(64, device="cuda")
    y = torch.zeros(64, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event(enable_timing=False) for _ in range(256)]
    evt = [torch.cuda.Event(enable_timing=True) for _ in range(256)]
    evb = [torch.cuda.Event(enable_timing=False, blocking=True) for _ in range(256)]
    big = torch.zeros(1 << 26, device="cuda")

    def kernels(n):
        with torch.cuda.stream(s1):
            for _ in range(n):
                x.add_(1)

    def kernels_rec(evl):
        def f(n):
            with torch.cuda.stream(s1):
                for i in range(n):
                    x.add_(1)
                    evl[i & 255].record(s1)
        return f

    def kernels_rec_wait(n):
        for i in range(n):
            with torch.cuda.stream(s1):
                x.add_(1)
                evs[i & 255].record(s1)
            s2.wait_event(evs[i & 255])
            with torch.cuda.stream(s2):
                y.add_(1)

    def long_kernels(n):  # ~0.3 ms each: the GPU, not the host, sets the pace
        with torch.cuda.stream(s1):
            for _ in range(n):
                big.add_(1)

    def long_kernels_rec_wait(n):
        for i in range(n):
            with torch.cuda.stream(s1):
                big.add_(1)
                evs[i & 255].record(s1)
            s2.wait_event(evs[i & 255])
            with torch.cuda.stream(s2):
                y.add_(1)

    kernels(2000)
    run("tiny kernels, one stream", 40000, kernels)
    run("+ event record (timing disabled) per kernel", 40000, kernels_rec(evs))
    run("+ event record (timing enabled) per kernel", 40000, kernels_rec(evt))
    run("+ event record (blocking sync) per kernel", 40000, kernels_rec(evb))
    run("+ record, other stream waits + launches", 20000, kernels_rec_wait)
    run("0.3 ms kernels, one stream", 3000, long_kernels)
    run("0.3 ms kernels + record + cross-stream wait + tiny kernel", 3000, long_kernels_rec_wait)


if __name__ == "__main__":
    main()
