# Round-3 profile bundle (run on the GPU box through gpurun): kernel-trace stats + separate PMC passes of the two bench workloads.
# Writes under gpurun_out/p3/; the summaries are copied into profiles/ as r03_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/p3; rm -rf $O; mkdir -p $O
# ---- frames workload (BASELINE configs[1]), one frame in flight: the conditions of the line's roofline object
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-extras --in-flight 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-extras > $O/kt3.log 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-extras --in-flight 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmcA -- $B > $O/pmcA.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmcB -- $B > $O/pmcB.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcC -- $B > $O/pmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcD -- $B > $O/pmcD.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcE -- $B > $O/pmcE.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $O/pmcF -- $B > $O/pmcF.log 2>&1
python scripts/pmc_summary.py $O/pmcA $O/pmcB $O/pmcC $O/pmcD $O/pmcE $O/pmcF > $O/pmc_all.json
# ---- rollout workload (BASELINE configs[4]) in fp32, one simulation in flight
R="python bench.py --workload rollout --steps 3 --warmup 1 --sims-per-gpu 2 --in-flight 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rkt -- $R > $O/rkt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/rpmcC -- $R > $O/rpmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/rpmcD -- $R > $O/rpmcD.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/rpmcA -- $R > $O/rpmcA.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/rpmcB -- $R > $O/rpmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/rpmcE -- $R > $O/rpmcE.log 2>&1
python scripts/pmc_summary.py $O/rpmcA $O/rpmcB $O/rpmcC $O/rpmcD $O/rpmcE > $O/rollout_pmc_all.json
python - <<'PY'
import json, glob, shutil
O = 'gpurun_out/p3'
for src, dst in (('pmc_all.json', 'pmc_ngp.json'), ('rollout_pmc_all.json', 'rollout_pmc_ngp.json')):
    d = json.load(open(f'{O}/{src}'))
    keep = {k: v for k, v in d.items() if 'ngp' in k or 'k_build' in k}
    json.dump(keep, open(f'{O}/{dst}', 'w'), indent=1)
    for k, v in keep.items():
        if any(t in k for t in ('render_iter', 'march_ahead', 'uniform_x16')):
            print(k[:70], v.get('launches'), {c: round(x['per_launch'], 1) for c, x in v.items() if isinstance(x, dict) and c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU', 'TCC_HIT_sum', 'TCC_REQ_sum')})
for name, out in (('kt', 'kernel_stats.csv'), ('kt3', 'kernel_stats_in_flight3.csv'), ('rkt', 'rollout_kernel_stats.csv')):
    shutil.copy(glob.glob(f'{O}/{name}/*/*kernel_stats.csv')[0], f'{O}/{out}')
PY
head -6 $O/kernel_stats.csv | cut -c1-160; head -4 $O/rollout_kernel_stats.csv | cut -c1-160
# drop the bulky raw traces from what travels back
rm -rf $O/kt $O/kt3 $O/rkt $O/pmc? $O/rpmc?
