"""Stall attribution per kernel family from the rocprofv3 --pmc passes of scripts/profile_stalls.sh:
    python scripts/pmc_stalls.py [--detail] [--min-us=20] <pass dir> ...
Every counter is averaged per launch over the dispatches of a family (kernel name without template arguments; with
--detail: name + template arguments + grid size, i.e. per layer shape).  Derived columns (units: SQ_WAVE_CYCLES,
SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over
SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs -- MI355X_MICROARCH.md, cycle-constants table and DVFS note):
  us          dispatch duration (End - Start timestamp of the counter pass that carries SQ_WAIT_ANY)
  GHz         GRBM_GUI_ACTIVE / 8 / us          (reads high on dispatches < 0.3 ms)
  mfma        SQ_VALU_MFMA_BUSY_CYCLES / (GUI/8 * 256 CUs * 4 SIMDs)
  wait        SQ_WAIT_ANY / SQ_WAVE_CYCLES      wave parked in s_waitcnt / s_barrier
  stall       SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES issue stall (MFMA dependency / pipe busy)
  issue       SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  valu vmem lds sca misc      SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES
  ldsstall    SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  bankc       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2hit       TCC_HIT / (TCC_HIT + TCC_MISS)
  rdlat       TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ     cycles per L1->L2 read request
  tcpstall    TCP_PENDING_STALL_CYCLES / (GUI/8 * 256)        share of CU-cycles the L1 stalls on pending misses
  occ         SQ_WAVE_CYCLES*4 / (GUI/8 * 1024 SIMDs)         resident waves per SIMD, time-averaged
  vmlvl       SQ_INST_LEVEL_VMEM / SQ_WAVE_CYCLES             vector-memory instructions in flight per wave (same pass)
"""
import collections, csv, glob, re, sys

detail = "--detail" in sys.argv
min_us = 20.0
dirs = []
for a in sys.argv[1:]:
    if a.startswith("--min-us="):
        min_us = float(a.split("=")[1])
    elif not a.startswith("--"):
        dirs.append(a)

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
regs = {}
seen = set()
for d in dirs:
    for f in glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k[:k.rfind("(")] if k.endswith(")") and "(" in k else k
            k = k.replace("void ", "")
            if detail:
                fam = f"{k} grid={r.get('Grid_Size', '?')}"
            else:
                fam = re.sub(r"<.*", "", k)
            c = r["Counter_Name"]
            a = agg[fam][c]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            regs[fam] = (r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("LDS_Block_Size", "?"), r.get("Workgroup_Size", "?"))
            if c == "SQ_WAIT_ANY" and "Start_Timestamp" in r:
                key = (f, r.get("Dispatch_Id"))
                if key not in seen:
                    seen.add(key)
                    dur[fam][0] += 1
                    dur[fam][1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3


def avg(fam, c):
    a = agg[fam].get(c)
    return a[1] / a[0] if a and a[0] else None


def ratio(x, y, scale=1.0):
    return None if x is None or not y else scale * x / y


def fmt(v, spec):
    return " " * (int(re.match(r"\d+", spec).group()) - 1) + "-" if v is None else format(v, spec)


rows = []
for fam in agg:
    us = dur[fam][1] / dur[fam][0] if dur[fam][0] else None
    n = max(a[0] for a in agg[fam].values())
    gui = avg(fam, "GRBM_GUI_ACTIVE")
    wc = avg(fam, "SQ_WAVE_CYCLES")
    cu_cyc = gui / 8.0 if gui else None
    tot = (us or 0) * (dur[fam][0] or 0)
    rows.append(dict(
        fam=fam, n=n, us=us, tot=tot,
        ghz=ratio(cu_cyc, us, 1e-3),
        mfma=ratio(avg(fam, "SQ_VALU_MFMA_BUSY_CYCLES"), cu_cyc * 1024 if cu_cyc else None),
        wait=ratio(avg(fam, "SQ_WAIT_ANY"), wc), stall=ratio(avg(fam, "SQ_WAIT_INST_ANY"), wc),
        issue=ratio(avg(fam, "SQ_ACTIVE_INST_ANY"), wc),
        valu=ratio(avg(fam, "SQ_ACTIVE_INST_VALU"), wc), vmem=ratio(avg(fam, "SQ_ACTIVE_INST_VMEM"), wc),
        lds=ratio(avg(fam, "SQ_ACTIVE_INST_LDS"), wc), sca=ratio(avg(fam, "SQ_ACTIVE_INST_SCA"), wc),
        misc=ratio(avg(fam, "SQ_ACTIVE_INST_MISC"), wc), ldsstall=ratio(avg(fam, "SQ_WAIT_INST_LDS"), wc),
        bankc=ratio(avg(fam, "SQ_LDS_BANK_CONFLICT"), avg(fam, "SQ_LDS_IDX_ACTIVE")),
        l2hit=ratio(avg(fam, "TCC_HIT_sum"), (avg(fam, "TCC_HIT_sum") or 0) + (avg(fam, "TCC_MISS_sum") or 0)),
        rdlat=ratio(avg(fam, "TCP_TCC_READ_REQ_LATENCY_sum"), avg(fam, "TCP_TCC_READ_REQ_sum")),
        tcpstall=ratio(avg(fam, "TCP_PENDING_STALL_CYCLES_sum"), cu_cyc * 256 if cu_cyc else None),
        occ=ratio(wc, cu_cyc * 1024 if cu_cyc else None, 4.0),
        vmlvl=ratio(avg(fam, "SQ_INST_LEVEL_VMEM"), wc),
    ))
rows = [r for r in rows if (r["us"] or 0) >= min_us or (detail and "mfma" in r["fam"] and (r["us"] or 0) >= 5)]
rows.sort(key=lambda r: -r["tot"])
cols = [("n", "6d"), ("us", "8.1f"), ("ghz", "5.2f"), ("occ", "5.2f"), ("mfma", "6.3f"), ("wait", "6.3f"), ("stall", "6.3f"),
        ("issue", "6.3f"), ("valu", "6.3f"), ("vmem", "6.3f"), ("lds", "6.3f"), ("sca", "6.3f"), ("misc", "6.3f"),
        ("ldsstall", "8.3f"), ("bankc", "6.3f"), ("l2hit", "6.3f"), ("rdlat", "7.0f"), ("tcpstall", "8.3f"), ("vmlvl", "6.2f")]
w = 96 if detail else 40
print(f"{'kernel':{w}s} " + " ".join(f"{c:>{int(re.match(r'[0-9]+', s).group())}s}" for c, s in cols) + "   vgpr/agpr/lds/wg")
for r in rows:
    print(f"{r['fam'][-w:]:{w}s} " + " ".join(fmt(r[c], s) for c, s in cols) + "   " + "/".join(regs[r["fam"]]))
print()
print("raw per-launch averages")
for r in rows:
    print(r["fam"])
    for c, (k, v) in sorted(agg[r["fam"]].items()):
        print(f"    {c:36s} {v / k:18.0f}  (x{k})")
