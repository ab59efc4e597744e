# rocprofv3 kernel-trace of the per-resolution table-gradient bench: per-launch durations of the binned scatter's two passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kt_gb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_gb -- python scripts/bench_grid_backward_levels.py ${1:-4194304} > gpurun_out/kt_gb.log 2>&1
python - <<'PY'
import csv, glob
tr = list(csv.DictReader(open(glob.glob('gpurun_out/kt_gb/*/*kernel_trace.csv')[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
for name in ('k_grid_bwd_binI', 'bin_reduce', 'backward_small', 'fillBuffer'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr if name in r['Kernel_Name']]
    print(name, len(d), [round(x) for x in d[3::4]])
PY
tail -11 gpurun_out/kt_gb.log
