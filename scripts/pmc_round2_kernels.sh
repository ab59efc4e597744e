# Counters of the kernels added or rebuilt in round 2, per kernel and launch: HBM-side traffic (FETCH_SIZE, WRITE_SIZE: KiB as
# reported -- the gfx950 x2 on FETCH_SIZE is applied by whoever reads them), L1 / L2 requests, VALU and LDS instruction counts.
#   scripts/uniform_rate.py      k_render_uniform_x16 with and without early stops (800x800, T = 512)
#   scripts/bench_operators.py   a full-frame training step: k_grid_bwd_bin<..>, k_grid_bwd_bin_reduce, k_ffmlp_bwd_fused<..>, k_sh_backward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/n2?
for tag in U O; do
  if [ $tag = U ]; then B="python scripts/uniform_rate.py"; else B="python scripts/bench_operators.py"; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/n2${tag}a -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/n2${tag}b -- $B > /dev/null 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/n2${tag}c -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/n2${tag}d -- $B > /dev/null 2>&1
done
python scripts/pmc_summary.py gpurun_out/n2Ua gpurun_out/n2Ub gpurun_out/n2Uc gpurun_out/n2Ud > gpurun_out/pmc_uniform.json
python scripts/pmc_summary.py gpurun_out/n2Oa gpurun_out/n2Ob gpurun_out/n2Oc gpurun_out/n2Od > gpurun_out/pmc_operators.json
python - <<'PY'
import json
out = {}
for f, keep in (("gpurun_out/pmc_uniform.json", ("k_render_uniform",)), ("gpurun_out/pmc_operators.json", ("k_grid_bwd", "k_grid_backward_small", "k_ffmlp_bwd", "k_ffmlp_forward", "k_sh_backward", "k_grid_forward"))):
    d = json.load(open(f))
    for k, v in d.items():
        if any(s in k for s in keep):
            out[k] = {c: (round(x["per_launch"], 1) if isinstance(x, dict) else x) for c, x in v.items()}
json.dump(out, open("gpurun_out/pmc_round2_kernels.json", "w"), indent=1)
for k, v in out.items():
    print(k[:60].ljust(62), v.get("launches"), "FETCH KiB", v.get("FETCH_SIZE"), "WRITE KiB", v.get("WRITE_SIZE"))
PY
