#!/bin/bash
# ablation builds of the ping-pong attention backward (compile-time switches) -> tools/bin/bw_<name>/liborv_mi355.so
cd /root/repo/orv_amd/csrc
for v in nodma_noy:"-DORV_BW_ABL_NODMA -DORV_BW_ABL_NOY" notr:"-DORV_BW_ABL_NODMA -DORV_BW_ABL_NOY -DORV_BW_ABL_NOTR" nolds:"-DORV_BW_ABL_NODMA -DORV_BW_ABL_NOY -DORV_BW_ABL_NOTR -DORV_BW_ABL_NOB128"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p ../../tools/bin/bw_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c attention_bwd.hip -o /tmp/attnb_$name.o || exit 1
  objs=$(ls build/*.o | grep -v attention_bwd.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/bw_$name/liborv_mi355.so $objs /tmp/attnb_$name.o
done
ls ../../tools/bin/ | grep bw_
