#!/bin/bash
# per-kernel register / LDS / occupancy table of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage)
# usage: bash tools/kres.sh deeplio_amd/csrc/bn_small.hip [filter]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DDLIO_HEADER_CRC=0u \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 | python3 -c '
import sys, re, subprocess
cur = None; rows = []
for l in sys.stdin:
    if "error" in l: print(l, end="")
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]}
        rows.append(cur); continue
    for k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs Spill", "VGPRs Spill"):
        m = re.search(re.escape(k) + r": (\d+)", l)
        if m and cur is not None and ("remark:     " + k + ":") in l: cur[k] = m.group(1)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-60s %5s %5s %5s %7s %4s %6s %6s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "occ", "lds", "sspill", "vspill"))
for r in rows:
    if flt in r["name"]:
        print("%-60s %5s %5s %5s %7s %4s %6s %6s %6s" % (r["name"][:60], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))
' "$2"
