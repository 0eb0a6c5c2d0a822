#!/usr/bin/env python3
"""HBM traffic per kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot
share a pass: TCC has 4 counter slots, MI355X_MICROARCH.md "PMC").

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o pmcf -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-isolated
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o pmcw -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-isolated
  python tools/pmc_traffic.py gpurun_out/pmc_f/pmcf_results.db gpurun_out/pmc_w/pmcw_results.db 2 profiles/rNN_pmc_traffic.json

Units: both counters are in KB.  gfx950 correction (same guide, "HBM"): FETCH_SIZE reports half of
the bytes of wide coalesced streaming reads -> corrected = 2*FETCH + WRITE; the uncorrected sum is
kept beside it.  Infinity-Cache hits are counted, so this is fabric traffic, an upper bound of DRAM
traffic."""
import json
import sqlite3
import sys

FAMILIES = {
    "conv2d_fwd_mfma": ("conv_fwd_kernel",),
    "conv3x3_bx3": ("conv3x3_bx3_", "fire_expand_fwd_kernel"),
    "conv2d_1x1": ("conv1x1_",),
    "conv2d_wgrad_mfma": ("conv_wgrad_kernel",),
    "wgrad3x3": ("wgrad3_kernel", "conv_wgrad_adirect"),
    "wgrad1x1": ("wgrad1x1_direct",),
    "wgrad_reduce": ("wgrad_reduce",),
    "batchnorm": ("chan_reduce", "chan_stats", "bn_apply", "bn_bwd", "bn_plane", "bn_coop_", "bn_small_", "bn_split16", "bn_pool_", "fire_stats_finalize", "bn_aff_"),
    "pool_se": ("maxpool", "gap_", "chan_scale", "pool3", "plane_dot"),
}


def collect(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, v in cur.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                  "where counter_name = ? group by kernel_name", (counter,)):
        out[name] = (n, v)
    return out


def main(db_f, db_w, steps, out_path):
    f, w = collect(db_f, "FETCH_SIZE"), collect(db_w, "WRITE_SIZE")
    res = {}

    def add(key, pred):
        nf = sum(n for k, (n, v) in f.items() if pred(k))
        vf = sum(v for k, (n, v) in f.items() if pred(k))
        vw = sum(v for k, (n, v) in w.items() if pred(k))
        res[key] = {"launches_%dsteps" % steps: nf, "FETCH_SIZE_KB_raw": vf, "WRITE_SIZE_KB": vw,
                    "hbm_bytes_per_step_corrected": (2 * vf + vw) * 1024 / steps,
                    "hbm_bytes_per_step_uncorrected": (vf + vw) * 1024 / steps}
    for fam, pats in FAMILIES.items():
        add(fam, lambda k, pats=pats: any(p in k for p in pats))
    add("all", lambda k: True)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
