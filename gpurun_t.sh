exec < /dev/null
mkdir -p gpurun_out/r03_tailprio
rm -f gpurun_out/r03_tailprio/ab.txt
for rep in 1 2 3 4; do
timeout 600 tools/ab_bench.sh r03_tailprio/ab.txt -r 1 -- plain MPVSS_TAIL_PRIORITY=0 -- boost MPVSS_TAIL_PRIORITY=1
done
