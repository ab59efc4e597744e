# rocprofv3 kernel-trace summary of scripts/bench_operators.py (every operator of a training step + the eval operators)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kt_ops
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_ops -- python scripts/bench_operators.py > gpurun_out/kt_ops.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt_ops/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r['Name'][:70].ljust(72), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6:9.3f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
