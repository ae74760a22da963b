cd tools/bin
for t in "2,256,256" "1,256,256"; do export ORV_GEMM_TILE=$t
for e in "X=1" "KB_ZERO=1" "KB_UNIT=1"; do echo "== tile $t $e"; for s in "4096 4096 4096 0" "8192 8192 8192 0"; do env $e timeout 60 ./kbench_gemm bench $s 20; done; done; done
