#!/bin/bash
# build variants of ONE source file into tools/bin/gv_<name>/liborv_mi355.so:  FILE=norm.hip VARIANTS="w6:-DORV_LN_WAVES=6 ..." bash tools/variants.sh
cd /root/repo/orv_amd/csrc
for v in $VARIANTS; do
  name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
  mkdir -p ../../tools/bin/gv_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c $FILE -o /tmp/var_$name.o || exit 1
  objs=$(ls build/*.o | grep -v "build/$FILE.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/gv_$name/liborv_mi355.so $objs /tmp/var_$name.o
done
ls ../../tools/bin/ | grep gv_ | tr '\n' ' '
