# kernel timeline of one Lego-like frame (BASELINE configs[3]: bound 1, cameras at r = 3.2): per-launch durations and gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/ktl && mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl -- python scripts/bench_lego_like.py 800 lego > gpurun_out/ktl.log 2>&1
python - <<'PY'
import csv, glob
tr = list(csv.DictReader(open(sorted(glob.glob('gpurun_out/ktl/*/*kernel_trace.csv'))[-1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'k_render_init' in r['Kernel_Name']]
fr = tr[idx[10]:idx[11]]          # frame number 10 (one at a time)
t0 = int(fr[0]['Start_Timestamp'])
prev_end = t0
busy = 0
for r in fr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][-40:]
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  grid {r.get('Grid_Size','?'):>9}  {name}")
    busy += e - s
    prev_end = e
t1 = max(int(r['End_Timestamp']) for r in fr)
print('frame span ms', (t1 - t0) / 1e6, 'busy ms', busy / 1e6, 'kernels', len(fr))
# time from the end of this frame to the start of the next one's first kernel
nxt = tr[idx[11]]
print('gap to next frame us', (int(nxt['Start_Timestamp']) - t1) / 1e3)
PY
