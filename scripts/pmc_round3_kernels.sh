# Counters of the kernels of a full-frame training step after round 3 (scripts/bench_operators.py), per kernel and launch: fabric-side
# traffic (FETCH_SIZE, WRITE_SIZE: KiB as reported -- the gfx950 x2 on FETCH_SIZE is applied by whoever reads them), L1 / L2 requests,
# VALU / LDS / MFMA instruction counts, wave cycles.  Separate --pmc passes, kernel trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/n3?
B="python scripts/bench_operators.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/n3a -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/n3b -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/n3c -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/n3d -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/n3e -- $B > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/n3a gpurun_out/n3b gpurun_out/n3c gpurun_out/n3d gpurun_out/n3e > gpurun_out/pmc_operators3.json
python - <<'PY'
import json
out = {}
d = json.load(open("gpurun_out/pmc_operators3.json"))
for k, v in d.items():
    if any(s in k for s in ("k_grid_", "k_ffmlp_", "k_ff_", "k_march_train", "k_composite_train")):
        out[k] = {c: (round(x["per_launch"], 1) if isinstance(x, dict) else x) for c, x in v.items()}
json.dump(out, open("gpurun_out/pmc_round3_kernels.json", "w"), indent=1)
for k, v in out.items():
    print(k[:70].ljust(72), v.get("launches"), "FETCH KiB", v.get("FETCH_SIZE"), "WRITE KiB", v.get("WRITE_SIZE"), "VALU", v.get("SQ_INSTS_VALU"), "SALU", v.get("SQ_INSTS_SALU"))
PY
