#!/bin/bash
# ablation builds of the ping-pong attention backward (compile-time switches, ONE stream removed at a time; wrong results) ->
# tools/bin/bw_<name>/liborv_mi355.so ; run with tools/bwd_abl_run.sh on the GPU box
cd /root/repo/orv_amd/csrc
for v in full:"" noy:"-DORV_BW_ABL_NOY" notr:"-DORV_BW_ABL_NOTR" nob128:"-DORV_BW_ABL_NOB128" nolds:"-DORV_BW_ABL_NOTR -DORV_BW_ABL_NOB128" nomfma:"-DORV_BW_ABL_NOMFMA" nodma:"-DORV_BW_ABL_NODMA" mfmaonly:"-DORV_BW_ABL_NODMA -DORV_BW_ABL_NOY -DORV_BW_ABL_NOTR -DORV_BW_ABL_NOB128"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p ../../tools/bin/bw_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -w $flags -c attention_bwd.hip -o /tmp/attnb_$name.o || exit 1
  objs=$(ls build/*.o | grep -v attention_bwd.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/bw_$name/liborv_mi355.so $objs /tmp/attnb_$name.o
done
ls ../../tools/bin/ | grep bw_
