exec < /dev/null
mkdir -p gpurun_out/r03_hash
rm -f gpurun_out/r03_hash/ab3.txt
for rep in 1 2 3; do
timeout 900 tools/ab_bench.sh r03_hash/ab3.txt -r 1 -- h8_d10 MPVSS_BENCH_DEPTH=10 -- h8_d8 MPVSS_BENCH_DEPTH=8 -- h8_d6 MPVSS_BENCH_DEPTH=6 -- h8_d5 MPVSS_BENCH_DEPTH=5
done
