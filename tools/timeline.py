#!/usr/bin/env python3
"""Per-stream timeline of the last training step in a rocprofv3 --kernel-trace database:
busy share and top kernels per millisecond and stream.  usage: timeline.py results.db"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    return re.sub(r"[<(].*", "", n)[:22]


def main(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, stream_id, start, end from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    # (the optimizer sweep may come as two launches per step: the tail bucket from inside backward + the rest at the end)
    per = max(1, round(len(adam) / max(1, sum(1 for r in rows if "pose_loss_fwd_kernel" in r[0]))))
    step = rows[adam[-1 - per] + 1:adam[-1] + 1]
    t0, t1 = step[0][2], step[-1][3]
    print("step wall %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
    ev = []
    for _, _, st, en in step:
        ev += [(st, 1), (en, -1)]
    ev.sort()
    c, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[c] += t - last
        last = t
        c += d
    print("ms with k kernels in flight:", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
    for sid in sorted({r[1] for r in step}):
        span = [r for r in step if r[1] == sid]
        print("stream %d: %d kernels, busy %.2f ms" % (sid, len(span), sum(r[3] - r[2] for r in span) / 1e6))
        for b in range(int((t1 - t0) / 1e6) + 1):
            lo, hi = b * 1e6, (b + 1) * 1e6
            busy, names = 0, collections.Counter()
            for n, _, st, en in span:
                a, e = max(st - t0, lo), min(en - t0, hi)
                if e > a:
                    busy += e - a
                    names[short(n)] += e - a
            if busy:
                print("  %2d ms: %3d%%  %s" % (b, busy / 1e4, ", ".join("%s %d" % (k, v / 1e4) for k, v in names.most_common(3))))


if __name__ == "__main__":
    main(sys.argv[1])
