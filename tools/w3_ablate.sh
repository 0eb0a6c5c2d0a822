#!/bin/bash
# timing ablations of conv_wgrad3.hip: builds tools/micro/_abl/lib_w3_<bits>.so (only that source is recompiled, the other
# objects are the current build's) here; on the GPU box `bash tools/w3_ablate.sh run` times tools/bench_wgrad3.py with each.
cd "$(dirname "$0")/.."
ABL=tools/micro/_abl; mkdir -p $ABL
if [ "$1" == "run" ]; then
  for f in $ABL/lib_w3_*.so; do echo "== $f"; python tools/variant_lib.py run $(basename $f .so | sed 's/^lib_//') -- python tools/bench_wgrad3.py | tail -7; done
  exit 0
fi
CRC=$(python -c "from deeplio_amd._header import abi_hash; print(abi_hash())")
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DDLIO_HEADER_CRC=${CRC}u -DW3_ABL=$b -c deeplio_amd/csrc/conv_wgrad3.hip -o /tmp/w3_abl_$b.o 2>/dev/null || exit 1
  objs=$(ls deeplio_amd/csrc/_obj/*.o | grep -v conv_wgrad3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ABL/lib_w3_$b.so $objs /tmp/w3_abl_$b.o && echo built $ABL/lib_w3_$b.so
done
