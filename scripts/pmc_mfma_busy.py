"""MFMA-busy summary per kernel family from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE:   python scripts/pmc_mfma_busy.py <dir>
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES summed over the SEs is not per-CU, so the ratio reported is
MFMA-busy cycles per CU-active cycle):  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
-- the share of SIMD-cycles in the dispatch's wall time during which the matrix pipe was busy; both raw counters are
printed so the ratio can be recomputed."""
import collections, csv, glob, sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            fam = k.split("<")[0].replace("void ", "")
            a = agg[fam][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
rows = []
for fam, cs in agg.items():
    mf = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])
    if mf[1] <= 0:
        continue
    n = mf[0]
    gui = cs.get("GRBM_GUI_ACTIVE", [0, 0.0])[1]
    busy = cs.get("SQ_BUSY_CYCLES", [0, 0.0])[1]
    wave = cs.get("SQ_WAVE_CYCLES", [0, 0.0])[1]
    simd_cycles = gui / 8.0 * 256 * 4 if gui else 0.0
    rows.append((mf[1], fam, n, mf[1] / n, busy / n, wave / n, gui / n, mf[1] / simd_cycles if simd_cycles else float("nan")))
print(f"{'kernel family':44s} {'launches':>8s} {'MFMA_BUSY/launch':>17s} {'SQ_BUSY/launch':>15s} {'WAVE_CYC/launch':>16s} {'GUI_ACTIVE/launch':>18s} {'mfma busy share':>16s}")
for _, fam, n, a, b, w, g, share in sorted(rows, reverse=True):
    print(f"{fam[-44:]:44s} {n:8d} {a:17.0f} {b:15.0f} {w:16.0f} {g:18.0f} {share:16.3f}")
