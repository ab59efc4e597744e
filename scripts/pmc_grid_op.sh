# Counters of the grid_encode_forward operator (k_grid_forward_g4) on 2 M random / ray-coherent points: L1 -> L2 request rate, L2 hit
# rate, fabric bytes.  Run on the GPU box; copy gpurun_out/grid_op_pmc.json into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/gop_*
for m in random coherent; do
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/gop_A_$m -- python scripts/bench_ops.py 2097152 $m > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/gop_B_$m -- python scripts/bench_ops.py 2097152 $m > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/gop_C_$m -- python scripts/bench_ops.py 2097152 $m > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gop_D_$m -- python scripts/bench_ops.py 2097152 $m > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, collections
out = {}
for m in ("random", "coherent"):
    for p in "ABCD":
        for f in glob.glob(f"gpurun_out/gop_{p}_{m}/*/*counter_collection.csv"):
            rows = [r for r in csv.DictReader(open(f)) if "k_grid_forward_g4" in r["Kernel_Name"]]
            # bench_ops.py times the plain table first (3 warm-up + 20 timed launches), then the same with records: split by dispatch order
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})
            half = ids[len(ids) // 2] if ids else 0
            for r in rows:
                variant = "records" if int(r["Dispatch_Id"]) >= half else "plain"
                key = f"{m}/{variant}"
                d = out.setdefault(key, collections.defaultdict(lambda: [0.0, set()]))
                d[r["Counter_Name"]][0] += float(r["Counter_Value"]); d[r["Counter_Name"]][1].add(r["Dispatch_Id"])
res = {k: {c: v[0] / max(1, len(v[1])) for c, v in d.items()} for k, d in out.items()}
B = 2097152
for k, d in res.items():
    if "TCP_TCC_READ_REQ_sum" in d:
        d["l1_miss_requests_per_point"] = d["TCP_TCC_READ_REQ_sum"] / B
        d["mean_l2_read_latency_cycles"] = d["TCP_TCC_READ_REQ_LATENCY_sum"] / max(1.0, d["TCP_TCC_READ_REQ_sum"])
    if "TCC_REQ_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_REQ_sum"])
    if "FETCH_SIZE" in d:
        d["fabric_bytes_per_point"] = d["FETCH_SIZE"] * 1024 * 2 / B      # KiB, gfx950 x2 (MI355X_MICROARCH.md)
json.dump(res, open("gpurun_out/grid_op_pmc.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v, 3) for c, v in d.items() if c[0].islower()})
PY
