# developer aid: first3x3_fwd_kernel per-launch time over workgroups per CU (run on the GPU box)
for g in 2 3 4 6 9; do
SRX_FIRST3_WGS_PER_CU=$g SRX_BENCH_SHAPES=gpurun_out/shapes_f3_$g.txt python bench.py --no-cpu-baseline --no-other-configs --no-dp-rehearsal --no-parity > gpurun_out/b_f3.log 2>&1
echo "wgs/cu $g: $(grep first3 gpurun_out/shapes_f3_$g.txt | head -1)"
done
