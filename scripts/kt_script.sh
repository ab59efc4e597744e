# rocprofv3 kernel-trace summary of any script: bash scripts/kt_script.sh <script.py> [args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kt_s
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_s -- python "$@" > gpurun_out/kt_s.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt_s/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r['Name'][:70].ljust(72), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.3f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
tail -2 gpurun_out/kt_s.log
