"""Turns a rocprofv3 kernel trace (…_kernel_trace.csv) into a small text report of the steady state:
per stream, which kernels ran when (last WINDOW ms of the run), how busy each stream was, and how long
no kernel at all was running.  Usage: python scripts/trace_timeline.py TRACE.csv OUT.txt [WINDOW_MS [MIN_KERNEL_MS]]"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "").replace("zkmi::", "")
    n = n.split("(")[0]
    n = re.sub(r"Fp28<(\w+)28Params\s*>", r"\g<1>28", n)
    n = re.sub(r"Fq2T<Fq28\s*>", "Fq2_28", n)
    return n.strip()


def main():
    path, out = sys.argv[1], sys.argv[2]
    window_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
    min_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
    ev = []
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r["Kernel_Name"])))
    tend = max(e[1] for e in ev)
    t0 = tend - int(window_ms * 1e6)
    win = sorted(e for e in ev if e[1] > t0)
    lines = ["steady-state window: last %.1f ms of %s" % (window_ms, path), ""]
    busy = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    for s, e, st, name in win:
        busy[st] += (e - max(s, t0)) / 1e6
        per_kernel[(st, name)][0] += 1
        per_kernel[(st, name)][1] += (e - max(s, t0)) / 1e6
    # time with no kernel running anywhere
    marks = sorted([(max(s, t0), 1) for s, e, _, _ in win] + [(e, -1) for s, e, _, _ in win])
    depth, last, idle, overlap = 0, t0, 0.0, collections.defaultdict(float)
    for t, d in marks:
        overlap[depth] += (t - last) / 1e6
        if depth == 0:
            idle += (t - last) / 1e6
        depth += d
        last = t
    lines.append("stream busy time (ms of %.1f): " % window_ms + ", ".join("s%s %.2f" % (k, v) for k, v in sorted(busy.items())))
    lines.append("no kernel running: %.2f ms; concurrent-kernel histogram (ms at depth d): " % idle
                 + ", ".join("%d:%.2f" % (k, v) for k, v in sorted(overlap.items())))
    n_acc = sum(c for (st, name), (c, _) in per_kernel.items() if name.startswith("k_accum_g1") or name.startswith("k_accum<Fq28"))
    lines.append("G1 accumulation launches in the window: %d (4 per proof -> %.2f ms per proof)" % (n_acc, 4 * window_ms / max(1, n_acc)))
    lines.append("")
    lines.append("per (stream, kernel): launches, total ms, average ms")
    for (st, name), (c, tot) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        lines.append("  s%-2s %-46s %4d %9.3f %8.3f" % (st, name[:46], c, tot, tot / c))
    lines.append("")
    lines.append("timeline (start ms, end ms, duration ms, stream, kernel); kernels shorter than %g ms omitted" % min_ms)
    for s, e, st, name in win:
        if (e - s) / 1e6 >= min_ms:
            lines.append("%9.3f %9.3f %8.3f s%-2s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, st, name[:60]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
