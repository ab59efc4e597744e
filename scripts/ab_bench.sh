#!/bin/bash
# A/B of the fused renderer's march variants: flags bit0 = no march-ahead queue, bit1 = no coarse occupancy filter
for f in 0 1 2 3; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --debug-flags $f 2>/dev/null | tail -1 > /tmp/ab_$f.json
  python - "$f" <<'PY'
import sys, json
f = sys.argv[1]; d = json.load(open(f"/tmp/ab_{f}.json"))
print("flags", f, "samples/s", d["value"], "ms/frame", d["ms_per_step"], "in-kernel", d["roofline"]["samples_per_s_in_kernel"])
PY
done
