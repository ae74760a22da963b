#!/bin/bash
# build gemm_d8 variants into tools/bin/dv_<name>/liborv_mi355.so (compile-time switches)
# usage: VARIANTS="noread:-DORV_D8_ABL_NOREAD nodma:-DORV_D8_ABL_NODMA" bash tools/d8_variants.sh
cd /root/repo/orv_amd/csrc
for v in $VARIANTS; do
  name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
  mkdir -p ../../tools/bin/dv_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -w $flags -c gemm_d8.hip -o /tmp/d8_$name.o || exit 1
  objs=$(ls build/*.o | grep -v gemm_d8.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/dv_$name/liborv_mi355.so $objs /tmp/d8_$name.o
done
ls ../../tools/bin/ | grep dv_
