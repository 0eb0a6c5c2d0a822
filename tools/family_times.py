#!/usr/bin/env python3
"""Group the kernels of a rocprofv3 --kernel-trace database (rocpd results.db) into the families of
the training step and print ms per optimizer step for each -- the table DESIGN.md and bench.py's
roofline.other quote.  Run the trace on `bench.py --serial` for exclusive (isolated) times.

    python tools/family_times.py <results.db> <steps in the trace> [out.md] [note...]
"""
import os
import re
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _trace_window import step_phase_start

FAMILIES = [
    ("conv3x3 fwd+dgrad (split-bf16 MFMA; fused Fire expand pair)", ("conv3x3_bx3_", "conv3x3_bf16", "fire_expand_fwd_kernel")),
    ("stem conv fwd (fp32 MFMA)", ("conv_fwd_kernel",)),
    ("conv1x1 fwd+dgrad", ("conv1x1_",)),
    ("conv3x3 wgrad", ("wgrad3_kernel", "conv_wgrad_adirect")),
    ("conv1x1 wgrad", ("wgrad1x1_",)),
    ("other wgrad + slab reduce", ("conv_wgrad_kernel", "wgrad_reduce_kernel")),
    ("BN fwd statistics", ("chan_reduce_kernel<0", "bn16_reduce_kernel<0", "fire_stats_finalize", "chan_stats_finalize")),
    ("BN fwd apply", ("bn_plane_apply_kernel", "bn_apply_kernel", "bn16_plane_apply", "bn_split16_kernel")),
    ("BN fwd streaming apply (statistics from the conv epilogue; + max-pool)", ("bn_aff_apply_kernel", "bn_aff_pool_kernel")),
    ("BN fwd, one launch (statistics + apply, one read)", ("bn_coop_fwd_kernel", "bn_small_fwd_kernel")),
    ("BN bwd, one launch (reductions + apply, one read)", ("bn_coop_bwd_kernel", "bn_small_bwd_kernel")),
    ("BN bwd reduce", ("chan_reduce_kernel<1", "chan_reduce_kernel<2", "bn16_reduce_kernel<1")),
    ("BN bwd apply", ("bn_plane_bwd_kernel", "bn_bwd_apply_kernel", "bn16_plane_bwd")),
    ("max-pool / SE scale", ("maxpool", "chan_scale", "gap_", "pool16_", "gap16_", "cast_", "plane_dot")),
    ("linear (RNN projections, SE fc, heads, soft fusion)", ("linear_", "col_sum_kernel", "act_bwd_kernel", "se_fc_", "pair_fuse_",
                                                            "soft_fusion_", "heads_")),
    ("LSTM/GRU recurrences (incl. the streamed layer's weight-streaming products)", ("lstm_", "gru_", "init_state", "gemv_", "slab_reduce")),
    ("optimizer + weight re-layout", ("adam_kernel", "sgd_kernel", "rmsprop", "adadelta", "prep_")),
]


def main(db, steps, out=None, note=""):
    cur = sqlite3.connect(db).cursor()
    t0, before = step_phase_start(cur)
    rows = cur.execute("select name, count(*), sum(end-start)/1e6 from kernels where name not like '%spin_kernel%' and start >= ? "
                       "group by name", (t0 if t0 is not None else -1,)).fetchall()
    if t0 is not None:
        note = (note + "; the %d launches in front of the first step (model construction) are left out" % before).lstrip("; ")
    fam = {n: [0, 0.0] for n, _ in FAMILIES}
    fam["everything else"] = [0, 0.0]
    for name, c, ms in rows:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", name)
        for fn, pats in FAMILIES:
            if any(p in name for p in pats):
                fam[fn][0] += c
                fam[fn][1] += ms
                break
        else:
            fam["everything else"][0] += c
            fam["everything else"][1] += ms
    tot = sum(v[1] for v in fam.values())
    lines = ["| family | launches/step | ms/step | % |", "|---|---|---|---|"]
    for n, (c, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %.1f | %.3f | %.1f |" % (n, c / steps, ms / steps, 100 * ms / tot))
    lines.append("| **total** | | **%.3f** | |" % (tot / steps))
    text = ("# kernel families, ms per optimizer step\n\n%s\n\n" % note) + "\n".join(lines) + "\n"
    print(text)
    if out:
        open(out, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None, " ".join(sys.argv[4:]))
