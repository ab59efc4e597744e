# kernel timeline of one frame, one at a time: per-launch durations and gaps.  `bash scripts/lego_trace.sh lego` = BASELINE configs[3]-like
# (bound 1, cameras at r = 3.2), `bash scripts/lego_trace.sh b2` = the bound-2 bench frame
WHICH=${1:-lego}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/ktl_$WHICH && mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl_$WHICH -- python scripts/bench_lego_like.py 800 $WHICH > gpurun_out/ktl_$WHICH.log 2>&1
python - $WHICH <<'PY'
import csv, glob, sys
tr = list(csv.DictReader(open(sorted(glob.glob(f'gpurun_out/ktl_{sys.argv[1]}/*/*kernel_trace.csv'))[-1])))
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
