"""Turns a rocprofv3 kernel trace (…_kernel_trace.csv) into a small text report of the STEADY STATE: per stream, which
kernels ran when, how busy each stream was, how long no kernel at all was running and how many kernels were resident.

The window is taken from the MIDDLE of the run: from the start of the 4th G1 accumulation launch to the end of the 4th from
last (a run of >= 16 proofs: warm-up, pipeline fill and drain stay outside).  Round 4's report took the LAST 60 ms of an
8-proof run -- the pipeline draining and then nothing -- and read "no kernel running: 35.7 ms" off it.
Usage: python scripts/trace_timeline.py TRACE.csv OUT.txt [WINDOW: mid | all | <last ms>] [MIN_KERNEL_MS] [LIST_MS]"""
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
    mode = sys.argv[3] if len(sys.argv) > 3 else "mid"
    min_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
    list_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 45.0  # the kernel-by-kernel listing covers this much of the window
    ev = []
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r["Kernel_Name"])))
    ev.sort()
    tbeg, tend = min(e[0] for e in ev), max(e[1] for e in ev)
    acc = [e for e in ev if e[3].startswith("k_accum_g1")]
    if mode == "mid" and len(acc) >= 9:
        t0, t1 = acc[3][0], acc[-4][1]
        what = "from the 4th G1 accumulation launch to the end of the 4th from last (%d such launches in the run)" % len(acc)
    elif mode in ("mid", "all"):
        t0, t1 = tbeg, tend
        what = "the whole run" + (" (fewer than 9 G1 accumulation launches: no middle to take)" if mode == "mid" else "")
    else:
        t0, t1 = tend - int(float(mode) * 1e6), tend
        what = "the last %s ms of the run" % mode
    window_ms = (t1 - t0) / 1e6
    # kernels clipped to the window
    win = sorted((max(s, t0), min(e, t1), st, name) for s, e, st, name in ev if e > t0 and s < t1)
    lines = ["window: %s = %.2f ms of %s" % (what, window_ms, path), ""]
    busy = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    for s, e, st, name in win:
        busy[st] += (e - s) / 1e6
        per_kernel[(st, name)][0] += 1
        per_kernel[(st, name)][1] += (e - s) / 1e6
    # time with no kernel running anywhere
    marks = sorted([(s, 1) for s, e, _, _ in win] + [(e, -1) for s, e, _, _ in win] + [(t1, 0)])
    depth, last, idle, overlap = 0, t0, 0.0, collections.defaultdict(float)
    for t, d in marks:
        overlap[depth] += (t - last) / 1e6
        if depth == 0:
            idle += (t - last) / 1e6
        depth += d
        last = t
    lines.append("stream busy time (ms of %.2f): " % window_ms + ", ".join("s%s %.2f" % (k, v) for k, v in sorted(busy.items())))
    lines.append("no kernel running: %.2f ms; concurrent-kernel histogram (ms at depth d): " % idle
                 + ", ".join("%d:%.2f" % (k, v) for k, v in sorted(overlap.items())))
    n_acc = sum(1 for s_, e_, st, name in win if name.startswith("k_accum_g1") and s_ >= t0)
    lines.append("G1 accumulation launches starting in the window: %d (4 per proof -> %.2f ms per proof)" % (n_acc, 4 * window_ms / max(1, n_acc)))
    lines.append("resident kernels: >= 1 for %.1f %% of the window, >= 3 for %.1f %%" % (
        100.0 * (1 - idle / window_ms), 100.0 * sum(v for k, v in overlap.items() if k >= 3) / window_ms))
    lines.append("")
    lines.append("per (stream, kernel): launches, total ms, average ms")
    for (st, name), (c, tot) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        lines.append("  s%-2s %-46s %4d %9.3f %8.3f" % (st, name[:46], c, tot, tot / c))
    lines.append("")
    lines.append("timeline of the first %.0f ms of the window (start ms, end ms, duration ms, stream, kernel); kernels shorter than %g ms omitted"
                 % (min(list_ms, window_ms), min_ms))
    for s, e, st, name in win:
        if (e - s) / 1e6 >= min_ms and (s - t0) / 1e6 <= list_ms:
            lines.append("%9.3f %9.3f %8.3f s%-2s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, st, name[:60]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
