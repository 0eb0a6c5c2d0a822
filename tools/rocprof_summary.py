#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database (results.db) into a per-kernel table
(count, total ms, avg us, share) -- the `--stats` view, written as markdown under profiles/."""
import os
import re
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _trace_window import step_phase_start


def main(db_path, out_path, steps, note=""):
    cur = sqlite3.connect(db_path).cursor()
    t0, before = step_phase_start(cur)
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels where name not like '%spin_kernel%' and start >= ? "
                       "group by name order by 3 desc", (t0 if t0 is not None else -1,)).fetchall()
    # (spin_kernel: the stream / hardware-queue probe of functional.assign_streams, once per process, not part of a step)
    tot = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n" % note)
        if t0 is not None:
            f.write("(the stepping phase of the trace: the %d launches in front of the first step -- model construction, parameter "
                    "initialisation, H2D copies -- are left out)\n\n" % before)
        f.write("total kernel time %.1f ms over %d steps (incl. warm-up) = %.1f ms/step\n\n" % (tot, steps, tot / steps))
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for n, c, s, a, mn, mx in rows:
            n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
            n = re.sub(r"\(.*", "", n)
            f.write("| `%s` | %d | %.2f | %.1f | %.1f | %.1f | %.1f |\n" % (n[:90], c, s, a, mn, mx, 100 * s / tot))
        # the bench's kernel families (ops.prof kinds): average launch duration to compare with
        # roofline.avg_launch_ms / roofline.isolated.avg_launch_ms of the bench JSON
        fam = {"conv3x3 split-bf16 MFMA (forward + data gradient)": ("conv3x3_bx3_kernel",),
               "conv2d_fwd_mfma, multi-tap on the fp32 MFMA (forward + data gradient)": ("conv_fwd_kernel",),
               "conv3x3 weight gradient (split-bf16 MFMA, wgrad3 kernel)": ("wgrad3_kernel", "conv_wgrad_adirect"),
               "conv1x1 weight gradient": ("wgrad1x1_direct",),
               "other weight gradients (staged fp32-MFMA kernel)": ("conv_wgrad_kernel",),
               "conv2d_1x1 (forward + data gradient)": ("conv1x1_",)}
        f.write("\n| bench family | kernel launches | total ms | avg us per kernel launch |\n|---|---|---|---|\n")
        for name, pats in fam.items():
            sel = [r for r in rows if any(p in r[0] for p in pats)]
            c, t = sum(r[1] for r in sel), sum(r[2] for r in sel)
            if c:
                f.write("| %s | %d | %.2f | %.1f |\n" % (name, c, t, 1e3 * t / c))
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), " ".join(sys.argv[4:]))
