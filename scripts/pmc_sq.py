"""Average SQ counters per kernel from rocprofv3 --pmc passes:  python scripts/pmc_sq.py <dir> [<dir> ...] [--filter substr]"""
import collections, csv, glob, sys
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for f in glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-48:]
            if flt and not any(x in r["Kernel_Name"] for x in flt):
                continue
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"    {c:32s} {v / n:16.0f}  (x{n})")
