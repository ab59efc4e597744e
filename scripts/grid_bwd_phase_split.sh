# Where the table-gradient scatter's time goes: scripts/bench_operators.py's grid_encode_backward line with parts of k_grid_bwd_bin
# switched off (NGP_GRID_BWD_SKIP, TIMING ONLY -- the gradients are wrong): 1 = no LDS merge table and no records, 2 = return after the
# corner loop (no sort, no records out, nothing for the second pass), 3 = both, 8 = no merging of same-cell neighbours in 16-lane rows.
mkdir -p gpurun_out/r03
for sk in 0 1 2 3 8 9; do
NGP_GRID_BWD_SKIP=$sk python scripts/bench_operators.py 2>/dev/null | grep -a "grid_encode_backward" | cut -c1-100 | sed "s/^/skip $sk: /"
done
