"""Writes profiles/INDEX.md: every file under profiles/ -> what it is, which build it was taken on, where DESIGN.md uses it.
    python scripts/profiles_index.py"""
import os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")

ROUND = {"r01": "round 1", "r02": "round 2", "r03": "round 3", "r04": "round 4", "r05": "round 5", "r06": "round 6"}
# (regex on the name without the round / build prefix, what, DESIGN.md place)
RULES = [
    (r"kernel_stats_(train|bench)_bs\d(_one_stream|_dp1)?\.csv$", "per-kernel totals of 8 training steps, `rocprofv3 --kernel-trace --stats` over `scripts/prof_train.py` (`_one_stream`: weight gradients in line, `LIDOG_BACKWARD_OVERLAP=0`)", "§5, §8"),
    (r"kernel_breakdown_train_bs\d(_one_stream|_dp1)?\.txt$", "the same trace summed per kernel family (`scripts/kernel_breakdown.py`)", "§8"),
    (r"step_timeline(_one_stream|_dp1)?\.txt$", "dispatch timeline of the last profiled step, per stream (`scripts/step_timeline.py`)", "§8"),
    (r"pmc_traffic_.*\.json$", "HBM bytes per launch of one kernel family: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes, FETCH doubled per MI355X_MICROARCH.md (`scripts/pmc_traffic.py`); read by `bench.py` for `roofline.traffic`", "§5"),
    (r"pmc_mfma_busy\.txt$", "MFMA-busy share per kernel family (`SQ_VALU_MFMA_BUSY_CYCLES` against `GRBM_GUI_ACTIVE`, `scripts/pmc_mfma_busy.py`)", "§8"),
    (r"pmc_sq_all_kernels\.txt$", "raw SQ counters per kernel (`scripts/pmc_sq.py`)", "§8"),
    (r"stalls_bs\d(_by_shape)?\.txt$", "stall attribution: wave cycles parked / stalled at issue / issuing, LDS bank conflicts, L2 hit rate, waves per SIMD per kernel family (`_by_shape`: per template instance and grid), 7 `--pmc` passes (`scripts/profile_stalls.sh`, `scripts/pmc_stalls.py`)", "§8 Round 6"),
    (r"clock_stamps\.txt$", "in-kernel shader clock of the matrix kernels (Δs_memtime / Δs_memrealtime per workgroup, diagnostic build `-DLIDOG_CLOCK_STAMP`, `scripts/clock_stamps.sh`)", "§8 Round 6"),
    (r"gemm_phases.*\.txt$", "`scripts/micro/gemm_phases.hip`: the gathered GEMM's loop with global loads / LDS stores / barriers switched off one by one, LDS-DMA variant", "§8 Round 6"),
    (r"gemm_units.*\.txt$|c1_source8k_gemm_units_ab\.txt$", "gathered GEMM alone, one unit per workgroup against several (`scripts/bench_gemm_units.py`) / C1 line with and without", "§8 Round 6"),
    (r"os_offset_group_split\.txt$", "output-stationary kernel with a tile's offsets cut into 3 groups, measured against the exact kernel and the two-pass path (experiment of commit d6c0076)", "§8 Round 6"),
    (r"gpu_test_durations.*\.txt$", "`pytest -m gpu --durations` of the round's suite", "§4"),
    (r"bench_default.*\.json$|bench\.json$|bench_line\.json$|bench\.log$|bench_dp1\.json$", "`python bench.py` (the headline line, with `roofline` and `cpu_baseline`)", "§5"),
    (r"bench_bs2\.json$", "`python bench.py --batch 2` (configs[2]'s per-GPU load)", "§5"),
    (r"bench_one_stream\.json$", "`LIDOG_BACKWARD_OVERLAP=0 python bench.py`", "§5"),
    (r"bench_single_rank_dp\.json$", "`LIDOG_BENCH_SINGLE_RANK_DP=1 python bench.py`: one-rank RCCL group with every data-parallel path on", "§6"),
    (r"bench_configs\.txt$|bench_c5\.json$", "secondary configs C1 / C4 / C5 (`scripts/bench_configs.py`)", "§5"),
    (r"^ab_|_ab_", "same-box alternating A/B of one experiment (bench lines or kernel times)", "§8 of that round"),
    (r"soak|flake|first_use|hipmemset|syncside", "first-use ticket bug of round 5: repetition counts, diagnosis, the `hipMemset` ordering probe, soak runs after the fix", "§8 Round 5"),
    (r"peer_probe|peer_allreduce", "statistics all-reduce latency through the peer one-shot path / RCCL / torch.distributed (two ranks on one GPU)", "§6"),
    (r"stats_tail", "latency of the in-kernel statistics finish (`scripts/bench_stats_tail.py`)", "§8 Round 5"),
    (r"bench_kernels|bench_os|bench_sorted|wgrad_|sweep", "kernel micro-benchmarks on the bench maps (`scripts/bench_kernels.py`, `bench_os.py`, `sweep_wgrad.py`)", "§8"),
    (r"experiments\.txt$", "numbered experiment log of that round", "§8"),
    (r"memcheck|host_", "host-side measurements (enqueue time per step, memory high-water marks)", "§3a"),
]


def describe(name):
    base = re.sub(r"^r0\d_([a-z]\d?_)?", "", name)
    for pat, what, where in RULES:
        if re.search(pat, base) or re.search(pat, name):
            return what, where
    return "see the DESIGN.md section of that round", "§8"


def build_tag(name):
    m = re.match(r"^(r0\d)_([a-z]\d?)_", name)
    if m and m.group(2) not in ("ab", "os", "c1", "c5"):
        return f"{m.group(1)}_{m.group(2)}"
    m = re.match(r"^(r0\d)_", name)
    return m.group(1) if m else "-"


def main():
    files = sorted(f for f in os.listdir(P) if f != "INDEX.md")
    out = ["# profiles/ — index", "",
           "Every figure DESIGN.md quotes comes from a file here.  `rNN` = the round; a letter behind it (`r05_f_…`) = the build of",
           "that round the set was taken on (`_f` = the round's final build).  Generated by `scripts/profiles_index.py`.", ""]
    for rnd in sorted(ROUND):
        rows = [f for f in files if f.startswith(rnd)]
        if not rows:
            continue
        out += [f"## {ROUND[rnd]} ({len(rows)} files)", "", "| file | build | what | DESIGN.md |", "|---|---|---|---|"]
        for f in rows:
            what, where = describe(f)
            out.append(f"| `{f}` | {build_tag(f)} | {what} | {where} |")
        out.append("")
    rest = [f for f in files if not re.match(r"^r0\d", f)]
    if rest:
        out += ["## other", ""] + [f"* `{f}`" for f in rest] + [""]
    open(os.path.join(P, "INDEX.md"), "w").write("\n".join(out))
    print(len(files), "files indexed")


if __name__ == "__main__":
    main()
