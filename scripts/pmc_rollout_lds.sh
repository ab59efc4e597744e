# LDS counters of the rollout workload's kernels (bench.py --workload rollout, fp32 and fp16): instructions, array cycles, conflict cycles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/rl?
for DT in f32 f16; do
R="python bench.py --workload rollout --steps 3 --warmup 1 --sims-per-gpu 2 --in-flight 1 --no-cpu-baseline --dtype $DT"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/rl_$DT -- $R > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/rl_$DT > gpurun_out/pmc_rollout_lds_$DT.json
done
python - <<'PY'
import json
for dt in ("f32", "f16"):
    d = json.load(open(f"gpurun_out/pmc_rollout_lds_{dt}.json"))
    for k, v in d.items():
        if "uniform" in k or "upsample" in k:
            print(dt, k[:70], {c: (round(x["per_launch"] / 1e6, 2) if isinstance(x, dict) else x) for c, x in v.items()})
PY
