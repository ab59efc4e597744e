cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python scripts/bench_ops.py 2097152 coherent"
rm -rf gpurun_out/opsA gpurun_out/opsB gpurun_out/opsC gpurun_out/opsF
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/opsA -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/opsB -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/opsC -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/opsF -- $B > /dev/null 2>&1
ls gpurun_out/opsA/*/ | head -2
