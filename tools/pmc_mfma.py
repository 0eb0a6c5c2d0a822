#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass of the training step.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY \
      SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_m -o pmcm -- \
      python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-isolated
  python tools/pmc_mfma.py gpurun_out/pmc_m/pmcm_results.db profiles/rNN_pmc_mfma.md

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the gfx94x derived-metric
formula the guide says rocprofv3 falls back to, with GRBM_GUI_ACTIVE divided by the 8 XCDs it is summed
over on this part (checked: GRBM_GUI_ACTIVE / 8 / dispatch duration = 2.25-2.9 GHz for every kernel; the
second column uses the dispatch duration x 2.4 GHz instead).  One v_mfma_f32_32x32x2_f32 keeps a SIMD busy for
64 cycles, so MfmaUtil x 157.3 TFLOP/s is the MFMA rate the kernel sustained INCLUDING the tile
padding it computes (channel counts that do not fill the 32-row tile) -- compare with the
algorithmic TFLOP/s of tools/conv_table.py.  The wave-state split (WAIT_ANY = parked on
s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing) is in quad-cycles and
given as a share of SQ_WAVE_CYCLES."""
import re
import sqlite3
import sys

def rate_label(k, util):
    """MFMA rate the utilisation stands for: fp32 MFMA 157.3 TF/s; three-piece bf16 split 2516.8 / 6 = 419.5 fp32-equivalent;
    two-piece fp16 split 2516.8 / 3 = 838.9 (the `..., true>` instantiations of the 3x3 / 1x1 split kernels)"""
    m = re.search(r"<(.*)>", k)
    a = [t.strip() for t in m.group(1).split(",")] if m else []
    name = k.split("<")[0]
    h2 = ((name.endswith("conv3x3_bx3_pc_kernel") and a[1:2] == ["true"]) or (name.endswith("fire_expand_fwd_kernel") and a[2:3] == ["true"])
          or (name.endswith("wgrad3_kernel") and a[3:4] == ["true"]) or (name.endswith("conv1x1_bx3_kernel") and a[2:3] == ["true"]))
    if h2:
        return "%.1f (fp32-equivalent: util x 838.9, two-piece)" % (util * 8.389)
    nat = (name.endswith("wgrad3_kernel") and a[2:3] == ["true"]) or (name.endswith("wgrad1x1_direct_kernel") and a[3:4] == ["true"]) or "bf16" in name
    if nat:
        return "%.1f (bf16 MFMA: util x 2516.8)" % (util * 25.168)
    bx3 = ("bx3" in name or name.endswith("fire_expand_fwd_kernel") or name.endswith("wgrad3_kernel")
           or (name.endswith("wgrad1x1_direct_kernel") and a[2:3] == ["true"])
           or (name.endswith("conv_wgrad_kernel") and a[7:8] == ["true"]))        # the stem weight gradient (round 6)
    if bx3:
        return "%.1f (fp32-equivalent: util x 419.5, three-piece)" % (util * 4.195)
    return "%.1f" % (util * 1.573)



def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    tab, dur = {}, {}
    for k, c, v, n, ns in rows:
        k = re.sub(r"\(anonymous namespace\)::|void ", "", k)
        k = re.sub(r"\(.*", "", k)
        tab.setdefault(k, {})[c] = (v, n)
        dur[k] = ns
    lines = []
    tot_busy = tot_act = 0.0
    for k, v in tab.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v:
            continue
        busy, n = v["SQ_VALU_MFMA_BUSY_CYCLES"]
        act = v["GRBM_GUI_ACTIVE"][0] / 8.0
        tot_act += act
        if busy <= 0:
            continue
        tot_busy += busy
        wc = v.get("SQ_WAVE_CYCLES", (0, 0))[0] or 1.0
        share = lambda c: 100.0 * v.get(c, (0, 0))[0] / wc
        lines.append((act, k, n, 100.0 * busy / (act * 256 * 4), 100.0 * busy / (dur[k] * 2.4 * 1024), share("SQ_WAIT_ANY"), share("SQ_WAIT_INST_ANY"),
                      share("SQ_ACTIVE_INST_ANY")))
    lines.sort(reverse=True)
    with open(out, "w") as f:
        f.write("# MFMA utilisation (rocprofv3 --pmc, kernels serialised by the counter pass)\n\n")
        f.write(__doc__.split("\n\n", 2)[2] + "\n\n")
        f.write("| kernel | launches | GPU-active cycles (M) | MfmaUtil %% | on duration x 2.4 GHz %% | = TFLOP/s incl. padding | WAIT_ANY %% | WAIT_INST_ANY %% | ACTIVE_INST %% |\n".replace("%%", "%"))
        f.write("|---|---|---|---|---|---|---|---|---|\n")
        for act, k, n, util, util_d, wa, wi, ai in lines:
            tf = rate_label(k, util)
            f.write("| `%s` | %d | %.1f | %.1f | %.1f | %s | %.0f | %.0f | %.0f |\n" % (k[:70], n, act / 1e6, util, util_d, tf, wa, wi, ai))
        f.write("\nAll kernels of the pass: MFMA busy %.1f %% of GPU-active cycles (MFMA kernels only: %.1f %%).\n"
                % (100.0 * tot_busy / (tot_act * 1024), 100.0 * tot_busy / (sum(l[0] for l in lines) * 1024)))
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
