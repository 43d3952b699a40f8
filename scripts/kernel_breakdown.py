"""Kernel time per step by category from a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME):
    python scripts/kernel_breakdown.py gpurun_out/prof/x_results.db STEPS [--csv out.csv]
Also reports the busy fraction of the main stream's span (gaps between dependent launches)."""
import collections, sqlite3, sys

db, steps = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
CATS = [("gathered GEMM (sconv fwd/dgrad)", ("k_sconv_gemm", "k_sconv_cin1<")),
        ("output-stationary 3^3 convolution", ("k_sconv_os",)), ("sparse wgrad", ("k_sconv_wgrad", "k_items_sum")),
        ("per-row reduction", ("k_sconv_reduce",)), ("conv2d (BEV head)", ("k_conv_", "k_pw_", "k_repack", "k_sum_splits", "k_sum_group", "k_support", "k_tile_lists", "k_group_lists")),
        ("BatchNorm / ReLU / add", ("k_colreduce", "k_sums_", "k_bn_", "k_partials", "k_stats_finish", "k_relu", "k_add")),
        ("coordinate + kernel maps", ("k_kernel_map", "k_pairs", "k_insert", "k_first_row", "k_stride", "scan_", "k_compact", "k_os_", "k_rows_", "k_bitmap", "DeviceRadixSort", "radix_sort", "rocprim")),
        ("BEV projection", ("k_bev",)), ("losses", ("k_dice",)), ("Adam", ("k_adam",)), ("weight transpose", ("k_transpose",))]
agg = collections.OrderedDict((n, [0, 0.0]) for n, _ in CATS)
agg["torch / runtime (copies, fills, cat, add)"] = [0, 0.0]
per_kernel = collections.defaultdict(lambda: [0, 0.0])
for name, s, e, _ in rows:
    short = name.split("(")[0]
    per_kernel[short][0] += 1
    per_kernel[short][1] += (e - s) / 1e6
    for cat, pats in CATS:
        if any(p in name for p in pats):
            agg[cat][0] += 1; agg[cat][1] += (e - s) / 1e6
            break
    else:
        agg["torch / runtime (copies, fills, cat, add)"][0] += 1
        agg["torch / runtime (copies, fills, cat, add)"][1] += (e - s) / 1e6
tot = sum(v[1] for v in agg.values())
print(f"{len(rows)} dispatches, {tot / steps:.2f} ms of kernel time per step ({steps} steps)")
for cat, (n, ms) in agg.items():
    print(f"  {cat:44s} {n / steps:7.1f} launches  {ms / steps:7.2f} ms  {100 * ms / tot:5.1f} %")
span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e6
print(f"span first->last kernel {span / steps:.2f} ms per step; busy fraction (all streams) {tot / span:.3f}")
if "--csv" in sys.argv:
    out = sys.argv[sys.argv.index("--csv") + 1]
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for k, (n, ms) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{n},{ms:.3f},{1e3 * ms / n:.2f},{100 * ms / tot:.2f}\n")
if "--top" in sys.argv:
    for k, (n, ms) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {k[-70:]:70s} {n / steps:7.1f}  {ms / steps:7.3f} ms  avg {1e3 * ms / n:8.1f} us")
