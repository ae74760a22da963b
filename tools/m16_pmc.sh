cd /root/repo; export LD_LIBRARY_PATH=/root/repo/orv_amd
ORV_ATTN_M16=1 FUSED=1 BOUND=12 ITERS=5 bash tools/pmc_kernel.sh m16 attn_fwd_m16 -- ./kbench_attn 4 > /dev/null 2>&1; echo "== m16"; cat gpurun_out/pmc_m16/summary.txt
ORV_ATTN_M16=0 FUSED=1 BOUND=12 ITERS=5 bash tools/pmc_kernel.sh pp32 attn_fwd_pp -- ./kbench_attn 4 > /dev/null 2>&1; echo "== pp (32x32x16)"; cat gpurun_out/pmc_pp32/summary.txt
find gpurun_out -name "*.csv" -size +200k -delete
