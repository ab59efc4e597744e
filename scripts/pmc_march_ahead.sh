# Counters of k_march_ahead on the bound-2 bench frame (scripts/bench_lego_like.py 800 b2): instruction mix and wave cycles per launch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ma?
B="python scripts/bench_lego_like.py 800 b2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/maa -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/mab -- $B > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/maa gpurun_out/mab > gpurun_out/pmc_march_ahead_all.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_march_ahead_all.json"))
for k, v in d.items():
    if "k_march_ahead" in k or "k_render_iter" in k:
        print(k[:60], {c: (round(x["per_launch"], 1) if isinstance(x, dict) else x) for c, x in v.items()})
PY
