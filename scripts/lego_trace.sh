cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/ktl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl -- python scripts/bench_lego_like.py > gpurun_out/ktl.log 2>&1
python - <<'PY'
import csv, glob
tr = list(csv.DictReader(open(sorted(glob.glob('gpurun_out/ktl/*/*kernel_trace.csv'))[-1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'k_render_init' in r['Kernel_Name']]
# frame number 10 of the first config (bound 1)
fr = tr[idx[10]:idx[11]]
for name in ('render_iter', 'render_compact'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in fr if name in r['Kernel_Name']]
    print(name, 'us', [round(x) for x in d])
t0=int(fr[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in fr)
print('frame span ms', (t1-t0)/1e6)
PY
