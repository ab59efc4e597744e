#!/usr/bin/env python3
"""Per-launch durations of the binned scatter's kernels from the latest rocprofv3 kernel trace under gpurun_out/<dir>."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt_ops"
f = max(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)
tr = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for name in sys.argv[2:] or ("k_grid_bwd_binI", "bin_reduce"):
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if name in r["Kernel_Name"]]
    print(name, len(t), [round(x) for x in t[-15:]])
