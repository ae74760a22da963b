#!/bin/bash
# build attention variants into tools/bin/av_<name>/liborv_mi355.so (compile-time switches), to be A/B'd with LD_LIBRARY_PATH
cd /root/repo/orv_amd/csrc
for v in base:"" noprio:"-DORV_PP_NOPRIO" $EXTRA_VARIANTS; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p ../../tools/bin/av_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c attention.hip -o /tmp/attn_$name.o || exit 1
  objs=$(ls build/*.o | grep -v attention.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/av_$name/liborv_mi355.so $objs /tmp/attn_$name.o
done
ls -la ../../tools/bin/av_*/
