# developer aid: PMC passes over the ESRGAN GAN step (tools/esrgan_step.py), averaged per dispatch for the kernels named in $1..
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_esr_$i -- python3 $R/tools/esrgan_step.py 2 > $R/gpurun_out/pmc_esr_$i.log 2>&1
  for k in "$@"; do echo "== $k"; python3 $R/tools/pmc_avg.py $R/gpurun_out/pmc_esr_$i "$k" || tail -3 $R/gpurun_out/pmc_esr_$i.log; done
done
