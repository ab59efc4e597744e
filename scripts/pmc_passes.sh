cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0"
rocprofv3 -L 2>/dev/null | grep -oE "(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TA_[A-Z_0-9a-z]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+)" | sort -u > gpurun_out/counters.txt
wc -l gpurun_out/counters.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmcA -- $B > gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmcB -- $B > gpurun_out/pmcB.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcC -- $B > gpurun_out/pmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcD -- $B > gpurun_out/pmcD.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcE -- $B > gpurun_out/pmcE.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/pmcF -- $B > gpurun_out/pmcF.log 2>&1
ls gpurun_out/pmc*/*/ | head -40
