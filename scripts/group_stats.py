"""Group a rocprofv3 kernel_stats CSV by kernel family:  python scripts/group_stats.py <csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms/step', round(tot / steps / 1e6, 2))
KEYS = [('k_sconv_gemm', 'gemm'), ('k_sconv_wgrad', 'wgrad'), ('k_items_sum', 'wgrad'), ('k_sconv_reduce', 'reduce'), ('k_stats_finish', 'reduce'),
        ('k_conv_', 'conv2d'), ('k_pw_', 'conv2d'), ('k_repack', 'conv2d'), ('k_sum_splits', 'conv2d'), ('k_bn', 'bn'), ('k_colreduce', 'bn'),
        ('k_partials_sum', 'bn'), ('k_bev', 'bev'), ('k_kernel_map', 'maps'), ('k_coords', 'maps'), ('k_hash', 'maps'), ('k_pairs', 'maps'),
        ('k_insert', 'maps'), ('k_stride', 'maps'), ('k_adam', 'adam'), ('k_relu', 'relu'), ('k_add', 'add'), ('k_transpose', 'transpose')]
def g(n):
    for key, lab in KEYS:
        if key in n: return lab
    return 'torch/other'
groups = {}
for r in rows:
    groups[g(r['Name'])] = groups.get(g(r['Name']), 0) + int(r['TotalDurationNs'])
for k, v in sorted(groups.items(), key=lambda x: -x[1]):
    print(f'{k:12s} {v / steps / 1e6:7.2f} ms/step {100 * v / tot:5.1f}%')
if len(sys.argv) > 3:
    for r in rows[:int(sys.argv[3])]:
        print(f"{r['Name'][:70]:70s} {int(r['Calls']) / steps:6.1f} {int(r['TotalDurationNs']) / steps / 1e6:7.3f}")
