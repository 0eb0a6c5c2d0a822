#!/usr/bin/env python3
"""VGPR / AGPR / scratch / occupancy / LDS of every kernel of a csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py conv_bx3.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "deeplio_amd", "csrc", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDLIO_HEADER_CRC=0u", "-c", src,
                    "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur, rows = None, []
for line in r.stderr.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): +(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split()[0]] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("%-60s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r_, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    n = re.sub(r"\(.*", "", n)
    if pat in n:
        print("%-60s %5s %5s %7s %4s %7s" % (n[:60], r_.get("VGPRs"), r_.get("AGPRs"), r_.get("ScratchSize"), r_.get("Occupancy"), r_.get("LDS")))
