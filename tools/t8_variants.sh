#!/bin/bash
# build gemm_t8 variants into tools/bin/gv_<name>/liborv_mi355.so (compile-time switches), A/B'd in the model with ORV_LIB=<path>
cd /root/repo/orv_amd/csrc
for v in $VARIANTS; do
  name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
  mkdir -p ../../tools/bin/gv_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c gemm_t8.hip -o /tmp/t8_$name.o || exit 1
  objs=$(ls build/*.o | grep -v gemm_t8.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/gv_$name/liborv_mi355.so $objs /tmp/t8_$name.o
done
ls ../../tools/bin/ | grep gv_
