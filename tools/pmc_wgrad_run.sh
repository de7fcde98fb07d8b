# (a pass with TA_TA_BUSY_sum / TA_BUFFER_WAVEFRONTS_sum / TA_*_STALLED_BY_TC_CYCLES_sum did not finish on this pool: left out)
# developer aid: PMC passes over the fp32 weight-gradient workloads of tools/pmc_workloads.py (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-wgrad256}
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${W}_$i -- python3 $R/tools/pmc_workloads.py $W > $R/gpurun_out/pmc_${W}_$i.log 2>&1
  python3 $R/tools/pmc_avg.py $R/gpurun_out/pmc_${W}_$i "wgrad_kernel" || tail -3 $R/gpurun_out/pmc_${W}_$i.log
done
