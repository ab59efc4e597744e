mkdir -p gpurun_out/r03
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03/ab_new_$i.json 2>/dev/null
NGP_NO_PRE_VERDICT=1 NGP_ITEM_SLOTS=0 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03/ab_old_$i.json 2>/dev/null
NGP_ITEM_SLOTS=0 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03/ab_pre_$i.json 2>/dev/null
done
python - <<'PY'
import json
for f in ("new_1","old_1","pre_1","new_2","old_2","pre_2"):
    d=json.loads(open(f"gpurun_out/r03/ab_{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]/1e9,3), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
