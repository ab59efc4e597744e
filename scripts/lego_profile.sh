# BASELINE configs[3] (Lego-like: bound 1, one cascade, cameras outside the box at r = 3.2, 800x800): rocprofv3 kernel-trace stats and
# the HBM / L2 counter passes of the fused renderer on that workload.  Run on the GPU box; copy gpurun_out/lego_* into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/lego_kt gpurun_out/lego_pmc?
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lego_kt -- python scripts/bench_lego_like.py 800 lego > gpurun_out/lego_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/lego_pmcC -- python scripts/bench_lego_like.py 800 lego > gpurun_out/lego_pmcC.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/lego_pmcD -- python scripts/bench_lego_like.py 800 lego > gpurun_out/lego_pmcD.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/lego_pmcB -- python scripts/bench_lego_like.py 800 lego > gpurun_out/lego_pmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/lego_pmcE -- python scripts/bench_lego_like.py 800 lego > gpurun_out/lego_pmcE.log 2>&1
python scripts/pmc_summary.py gpurun_out/lego_pmcB gpurun_out/lego_pmcC gpurun_out/lego_pmcD gpurun_out/lego_pmcE > gpurun_out/lego_pmc_all.json
python - <<'PY'
import json, glob, shutil
d = json.load(open('gpurun_out/lego_pmc_all.json'))
json.dump({k: v for k, v in d.items() if 'ngp' in k}, open('gpurun_out/lego_pmc.json', 'w'), indent=1)
shutil.copy(glob.glob('gpurun_out/lego_kt/*/*kernel_stats.csv')[0], 'gpurun_out/lego_kernel_stats.csv')
PY
grep '^{' gpurun_out/lego_bench.log > gpurun_out/lego_bench.jsonl
head -6 gpurun_out/lego_kernel_stats.csv | cut -c1-160
cat gpurun_out/lego_bench.jsonl
