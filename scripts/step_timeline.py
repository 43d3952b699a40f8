"""Per-dispatch timeline of the LAST training step in a rocprofv3 kernel-trace database:
    python scripts/step_timeline.py <x_results.db> [--gaps]
One line per dispatch in start order: start offset (us), duration (us), gap to the previous dispatch's end on the same
stream (us), stream, grid, kernel.  The step starts at the first `k_insert`-less forward kernel after the previous
k_adam.  Summary: per-stream busy time, sum of gaps on the main stream (host dispatch / dependency bubbles)."""
import sqlite3, sys

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
wx = "workgroup_size_x" if "workgroup_size_x" in cols else None
sel = "name, start, end, stream_id" + (f", {gx}" if gx else ", 0") + (f", {wx}" if wx else ", 0")
rows = c.execute(f"select {sel} from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "k_adam" in r[0] or "k_sgd" in r[0]]
if len(adam) < 2:
    sys.exit("need at least two optimiser steps in the trace")
lo, hi = adam[-2] + 1, adam[-1] + 1
step = rows[lo:hi]
t0 = step[0][1]
main = max(set(r[3] for r in step), key=lambda s: sum(1 for r in step if r[3] == s))
last_end = {}
busy, gaps = {}, 0.0
print(f"# last step: {len(step)} dispatches, {(step[-1][2] - t0) / 1e3:.1f} us wall; main stream = {main}")
print(f"{'start_us':>10s} {'dur_us':>9s} {'gap_us':>8s} {'strm':>5s} {'blocks':>8s}  kernel")
for name, s, e, st, g, w in step:
    gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
    last_end[st] = max(e, last_end.get(st, 0))
    busy[st] = busy.get(st, 0.0) + (e - s) / 1e3
    if st == main and gap > 0:
        gaps += gap
    blocks = (g // w) if (g and w) else 0
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:8.1f} {st:5d} {blocks:8d}  {name.split('(')[0][-60:]}")
print("# busy per stream (us):", {k: round(v, 1) for k, v in busy.items()})
print(f"# gaps on the main stream: {gaps:.1f} us")
