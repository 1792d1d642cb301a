exec < /dev/null
mkdir -p gpurun_out/r03_hash
rm -f gpurun_out/r03_hash/ab.txt
for rep in 1 2 3; do
timeout 600 tools/ab_bench.sh r03_hash/ab.txt -r 1 -- h6 MPVSS_BENCH_HASH_THREADS=6 -- h12 MPVSS_BENCH_HASH_THREADS=12 -- h12_d16 MPVSS_BENCH_HASH_THREADS=12 MPVSS_BENCH_DEPTH=16 -- h8_d10 MPVSS_BENCH_HASH_THREADS=8 MPVSS_BENCH_DEPTH=10
done
