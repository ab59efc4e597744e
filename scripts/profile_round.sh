# Round profile bundle (run on the GPU box through gpurun): kernel-trace stats + PMC passes of bench.py, operator microbenchmarks.
# Writes everything under gpurun_out/; copy the summaries you want judged into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt gpurun_out/pmc? gpurun_out/pmc_summary.json
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 --batched-views 0 --in-flight 1"   # one frame in flight: the conditions of the roofline leg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- $B > gpurun_out/kt.log 2>&1
# the default command (two frames in flight: kernels of the two streams overlap, per-launch durations grow)
rm -rf gpurun_out/kt2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt2 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --batched-views 0 > gpurun_out/kt2.log 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --batched-views 0 --in-flight 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmcA -- $B > gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmcB -- $B > gpurun_out/pmcB.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcC -- $B > gpurun_out/pmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcD -- $B > gpurun_out/pmcD.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcE -- $B > gpurun_out/pmcE.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/pmcF -- $B > gpurun_out/pmcF.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC gpurun_out/pmcD gpurun_out/pmcE gpurun_out/pmcF > gpurun_out/pmc_summary.json
python - <<'PY'
import json, glob, shutil
d = json.load(open('gpurun_out/pmc_summary.json'))
keep = {k: v for k, v in d.items() if 'ngp' in k}
json.dump(keep, open('gpurun_out/pmc_ngp.json', 'w'), indent=1)
for k, v in keep.items():
    if 'render_iter' in k:
        print(k[:40], {c: round(x['per_launch'], 1) for c, x in v.items() if isinstance(x, dict)})
shutil.copy(glob.glob('gpurun_out/kt/*/*kernel_stats.csv')[0], 'gpurun_out/kernel_stats.csv')
PY
head -5 gpurun_out/kernel_stats.csv | cut -c1-150
for m in random coherent; do python scripts/bench_ops.py 2097152 $m 2>/dev/null | grep '^{' ; done > gpurun_out/bench_ops.jsonl
cat gpurun_out/bench_ops.jsonl | cut -c1-220
python scripts/bench_run_path.py 2>/dev/null | tail -3
