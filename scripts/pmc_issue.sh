# One PMC pass answering "which issue port is busy": per-type active/issue cycles of k_render_iter on the bench workload.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0"
rm -rf gpurun_out/pmcG gpurun_out/pmcH
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcG -- $B > gpurun_out/pmcG.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace --output-format csv -d gpurun_out/pmcH -- $B > gpurun_out/pmcH.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmcG gpurun_out/pmcH > gpurun_out/pmc_issue.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_issue.json'))
for k, v in d.items():
    if 'render_iter' in k:
        for c, x in sorted(v.items()):
            if isinstance(x, dict): print(f"{c:28s} {x['sum']:.4g}")
PY
tail -3 gpurun_out/pmcG.log gpurun_out/pmcH.log
