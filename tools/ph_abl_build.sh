#!/bin/bash
# builds ablation variants of the library (phased GEMM without DMA / without MFMA) into tools/bin/abl_ph/<VARIANT>/
cd /root/repo/orv_amd/csrc
for v in NODMA NOMFMA; do
  mkdir -p ../../tools/bin/abl_ph/$v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DORV_PH_ABLATE_$v -c gemm.hip -o /tmp/gemm_$v.o &
done
wait
for v in NODMA NOMFMA; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/abl_ph/$v/liborv_mi355.so build/lib.cpp.o /tmp/gemm_$v.o build/attention.hip.o build/attention_bwd.hip.o build/norm.hip.o build/embed.hip.o build/backward.hip.o
done
ls -la ../../tools/bin/abl_ph/*/
