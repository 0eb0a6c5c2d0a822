#!/usr/bin/env python3
"""every kernel of the last training step of a rocprofv3 --kernel-trace database: start (us since the step
began), duration (us), stream, name -- for reading the serial parts of the step.  usage: step_dump.py results.db [lo_ms hi_ms]"""
import re
import sqlite3
import sys


def main(db, lo=None, hi=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, stream_id, start, end from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    # (the optimizer sweep may come as two launches per step: the tail bucket from inside backward + the rest at the end)
    per = max(1, round(len(adam) / max(1, sum(1 for r in rows if "pose_loss_fwd_kernel" in r[0]))))
    step = rows[adam[-1 - per] + 1:adam[-1] + 1]
    t0 = step[0][2]
    prev_end = {}
    for n, sid, st, en in step:
        a = (st - t0) / 1e3
        if lo is not None and not (lo * 1e3 <= a <= hi * 1e3):
            continue
        n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
        n = re.sub(r"\(.*", "", n)[:60]
        print("%9.1f %8.1f  s%-2d %s" % (a, (en - st) / 1e3, sid, n))


if __name__ == "__main__":
    main(sys.argv[1], *(float(v) for v in sys.argv[2:4]))
