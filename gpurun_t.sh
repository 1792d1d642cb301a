exec < /dev/null
mkdir -p gpurun_out/r03_c5
rm -f gpurun_out/r03_c5/loop_*.txt
for i in 1 2 3 4 5 6 7 8; do
timeout 300 python -m pytest tests/test_gpu_bench_multirank.py -x -q -m gpu -k "c5 or four" 2>&1 | tail -80 | cut -c1-900 > gpurun_out/r03_c5/loop_$i.txt
tail -1 gpurun_out/r03_c5/loop_$i.txt
done
