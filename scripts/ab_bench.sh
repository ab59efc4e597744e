#!/bin/bash
# A/B of fused-renderer variants through bench.py --debug-flags (see ngp_debug_disable_march_queue); usage: ab_bench.sh "0 2 4"
for f in ${1:-0 2 4}; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --debug-flags $f 2>/dev/null | tail -1 > /tmp/ab_$f.json
  python - "$f" <<'PY'
import sys, json
f = sys.argv[1]; d = json.load(open(f"/tmp/ab_{f}.json"))
print("flags", f, "samples/s", d["value"], "ms/frame", d["ms_per_step"], "in-kernel", d["roofline"]["samples_per_s_in_kernel"])
PY
done
