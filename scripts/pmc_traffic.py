"""Per-launch HBM traffic of the dominant kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE k_sconv_gemm > profiles/rNN_pmc_traffic.json
Units and the gfx950 correction follow MI355X_MICROARCH.md section HBM: both counters are in KiB; FETCH_SIZE tallies
128-B requests at 64 B, so read bytes = 2 * FETCH_SIZE * 1024 for wide coalesced reads (upper bound for our
128-B-line gathers); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fd, wd, pat = sys.argv[1], sys.argv[2], sys.argv[3]
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    launches = sum(n for k, (n, v) in fe.items() if pat in k)
    fetch_kib = sum(v for k, (n, v) in fe.items() if pat in k)
    write_kib = sum(v for k, (n, v) in wr.items() if pat in k)
    out = {"kernel": pat, "launches": launches, "fetch_kib_raw": fetch_kib, "write_kib": write_kib,
           "read_bytes_per_launch_corrected": 2 * fetch_kib * 1024 / launches,
           "read_bytes_per_launch_uncorrected": fetch_kib * 1024 / launches,
           "write_bytes_per_launch": write_kib * 1024 / launches,
           "traffic_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024 / launches,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `STEPS=2 python3 scripts/prof_train.py` (scripts/profile_round.sh), one stream; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
