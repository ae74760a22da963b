#!/bin/bash
# Control A/B: the guide's literal 256x256 8-phase template vs gemm_ph_kernel<256,0>, one process, interleaved rounds; then PMC
# (separate --pmc passes) of both kernels at 4096^3 random.   -> gpurun_out/tpl_ab.txt, gpurun_out/pmc_tpl_*/summary.txt
cd /root/repo
mkdir -p gpurun_out
( cd tools/bin && timeout 600 ./probe_gemm_template ${1:-5} ) > gpurun_out/tpl_ab.txt 2>&1
tail -40 gpurun_out/tpl_ab.txt
export TPL_PMC=1
bash tools/pmc_kernel.sh tpl_template gemm_tpl_kernel -- ./probe_gemm_template 1 > /dev/null 2>&1
bash tools/pmc_kernel.sh tpl_ph gemm_ph_kernel -- ./probe_gemm_template 1 > /dev/null 2>&1
for t in tpl_template tpl_ph; do echo "== $t"; cat gpurun_out/pmc_$t/summary.txt; done
