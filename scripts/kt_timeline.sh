# rocprofv3 kernel-trace of bench.py with ONE frame in flight: per-kernel totals and the last frame's launch timeline (durations and gaps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/kt1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-extras --in-flight 1 ${BENCH_EXTRA} > gpurun_out/kt1.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt1/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:10]:
    print(r['Name'][:60].ljust(62), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6:9.3f} ms", f"{float(r['AverageNs'])/1e3:8.1f} us", r['Percentage'])
tr = list(csv.DictReader(open(glob.glob('gpurun_out/kt1/*/*kernel_trace.csv')[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'k_get_rays' in r['Kernel_Name']]
fr = tr[idx[-2]:idx[-1]] if len(idx) > 1 else tr[idx[-1]:]
t0 = int(fr[0]['Start_Timestamp']); t1 = int(fr[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in fr)
print('one frame: span ms', (t1 - t0) / 1e6, 'busy ms', busy / 1e6, 'kernels', len(fr))
prev = None
line = []
for r in fr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    prev = e
    n = r['Kernel_Name']
    short = 'march' if 'march_ahead' in n else 'iter' if 'render_iter' in n else 'compact' if 'render_compact' in n else n.split('(')[0].split('::')[-1][:18]
    line.append(f"{short}:{(e - s) / 1e3:.0f}(+{gap:.0f})")
print(' '.join(line))
for name in ('march_ahead', 'render_iter', 'render_compact'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in fr if name in r['Kernel_Name']]
    print(name, 'total us', round(sum(d)), 'n', len(d))
PY
