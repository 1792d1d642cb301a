exec < /dev/null
mkdir -p gpurun_out/r03_final
timeout 600 tools/ab_bench.sh r03_final/k100.txt -k 100 -- k100 MPVSS_X=0
timeout 600 tools/ab_bench.sh r03_final/k100.txt -k 20 -- k20 MPVSS_X=0
