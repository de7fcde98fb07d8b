mkdir -p gpurun_out
for k in srgan esrgan; do
  SRX_NO_OVERLAP=1 timeout -k 10 300 python tools/overlap_check.py $k 5 > gpurun_out/ov_${k}_a.txt 2>&1
  timeout -k 10 300 python tools/overlap_check.py $k 5 > gpurun_out/ov_${k}_b.txt 2>&1
  if cmp -s gpurun_out/ov_${k}_a.txt gpurun_out/ov_${k}_b.txt; then echo "$k: identical"; else echo "$k: DIFFERENT"; diff gpurun_out/ov_${k}_a.txt gpurun_out/ov_${k}_b.txt | head -20; fi
done
for rep in 1 2; do
  for no in 1 0; do
    if [ $no = 1 ]; then export SRX_NO_OVERLAP=1; else unset SRX_NO_OVERLAP; fi
    echo "NO_OVERLAP=$no: $(timeout -k 10 300 python tools/srgan_step.py 40 2>&1 | tail -1)  $(timeout -k 10 300 python tools/esrgan_step.py 40 2>&1 | tail -1)"
  done
done
