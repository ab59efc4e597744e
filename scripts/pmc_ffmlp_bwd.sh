# LDS counters of the FFMLP kernels in a full-frame training step (scripts/bench_operators.py): instructions, array cycles, conflict cycles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fb?
B="python scripts/bench_operators.py"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/fba -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/fbb -- $B > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/fba gpurun_out/fbb > gpurun_out/pmc_ffmlp_all.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_ffmlp_all.json"))
out = {k: {c: (round(x["per_launch"], 1) if isinstance(x, dict) else x) for c, x in v.items()} for k, v in d.items() if "ffmlp" in k}
json.dump(out, open("gpurun_out/pmc_ffmlp.json", "w"), indent=1)
for k, v in out.items(): print(k[:64], v)
PY
