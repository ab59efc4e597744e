# rocprofv3 kernel-trace summary of bench.py: per-kernel totals and the last frame's iteration timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > gpurun_out/kt.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:6]:
    print(r['Name'][:50].ljust(52), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6:9.3f} ms", f"{float(r['AverageNs'])/1e3:8.1f} us", r['Percentage'])
tr = list(csv.DictReader(open(glob.glob('gpurun_out/kt/*/*kernel_trace.csv')[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'k_render_init' in r['Kernel_Name']]
fr = tr[idx[-1]:]
end = max(i for i, r in enumerate(fr) if 'render_compact' in r['Kernel_Name'] or 'render_iter' in r['Kernel_Name'])
fr = fr[:end + 1]
t0 = int(fr[0]['Start_Timestamp']); t1 = int(fr[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in fr)
print('last frame loop span ms', (t1 - t0) / 1e6, 'busy ms', busy / 1e6, 'kernels', len(fr))
for name in ('render_iter', 'render_compact'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in fr if name in r['Kernel_Name']]
    print(name, 'us', [round(x) for x in d])
PY
