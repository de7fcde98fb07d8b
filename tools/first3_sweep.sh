for g in 0 1 2 3; do
SRX_F3_DBG=$g SRX_BENCH_SHAPES=gpurun_out/shapes_f3d_$g.txt python bench.py --no-cpu-baseline --no-other-configs --no-dp-rehearsal --no-parity > gpurun_out/b_f3.log 2>&1
echo "dbg $g: $(grep first3 gpurun_out/shapes_f3d_$g.txt | head -1)"
done
