"""Headline benchmark: LiDAR scans/sec of the LiDOG training step (MinkUNet34 + BEV head) on synthetic
120k-point / 0.05 m scans (BASELINE.json configs[1], batch 4 per GPU), one process per GPU over RCCL.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward (is_train=True, BEV head included) + both DICE losses + backward + gradient
all-reduce + Adam update; inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# Five HIP streams of this library are busy in a data-parallel step (compute, weight gradients, next batch's
# coordinate maps, gradient buckets, plus torch's and RCCL's internal ones for three communicators); streams that
# land on the same hardware queue serialise.  Measured on one rank with every collective active: 8 queues 67.7 ms per
# step (the two extra communicators pushed the weight-gradient stream onto the compute stream's queue), 16 queues
# 50.7 ms; the runtime's default of 4 already serialised the weight gradients in round 1.  Must be set before the HIP
# runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# the host driver of this pool only supports dmabuf IPC (RCCL's peer buffers); normally already exported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # f32-input MFMA peak = vector fp32 peak (MI355X_MICROARCH.md, Matrix cores)
# scripts/profile_round.sh <tag> writes profiles/<tag>_pmc_traffic_sconv_gemm_mfma.json (FETCH_SIZE / WRITE_SIZE passes)
PMC_TRAFFIC_TAG = "r06_f"
PMC_TRAFFIC_FILE = f"{PMC_TRAFFIC_TAG}_pmc_traffic_sconv_gemm_mfma.json"
# An N > 1 run that prints nothing for this long is taken to be hung (a collective that never completes): the driver
# allows the whole command 600 s, so the silence limit plus one safe-mode re-run must fit well inside that.  Every rank
# reports its progress on stderr (beat() below) at least once per block of timed steps.
WATCHDOG_DEFAULT_S = 150.0
_T0 = time.time()


def beat(msg):
    """one progress line on stderr -- what the silence watchdogs above this process (supervise_rank, launch_ranks)
    listen for; never on stdout, which carries rank 0's JSON line only"""
    sys.stderr.write(f"bench.py[rank {os.environ.get('RANK', '0')} +{time.time() - _T0:.1f}s]: {msg}\n")
    sys.stderr.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="scans per GPU (configs[1]: bs=4)")
    ap.add_argument("--config", default="kitti120k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="build coordinate maps inside the forward pass")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--min-seconds", type=float, default=5.0,
                    help="repeat the timed block of --steps steps until this much step time is measured; the median block is reported")
    ap.add_argument("--max-blocks", type=int, default=9)
    return ap.parse_args()


class GemmTimer:
    """HIP events around every launch of the dominant kernel (the gathered GEMM, csrc/sconv_mfma.hip), on the
    stream it is launched on, with its algorithmic bytes/FLOPs."""

    def __init__(self):
        self.records = []
        self._enabled = False
        self.native = [0.0, 0.0, 0.0, 0.0]   # launches, ms, FLOPs, bytes timed inside the trunk executor

    @property
    def enabled(self):
        return self._enabled

    @enabled.setter
    def enabled(self, on):
        # the training step's convolutions are launched by the trunk executor (csrc/trunk.hip): its launches of the
        # same kernel are bracketed by events in C, on the stream they run on (lidog_trunk_gemm_timing)
        if bool(on) != self._enabled:
            from lidog_amd import _lib
            _lib.load().lidog_trunk_gemm_timing(1 if on else 0)
        self._enabled = bool(on)

    def wrap(self, me):
        orig = me._gemm
        timer = self

        def timed(A, gather, B, bias, m, Cin, Cout, out, scatter, tiles=None):
            if not timer.enabled:
                return orig(A, gather, B, bias, m, Cin, Cout, out, scatter, tiles)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(A, gather, B, bias, m, Cin, Cout, out, scatter, tiles)
            e1.record()
            rows = m.P if tiles is None else tiles[2]   # pairs this launch multiplies (the centre segment may be left out)
            # algorithmic traffic of one launch: every distinct input row and weight read once, every
            # product row written once, the gather index read once (SURVEY.md 8(d) per-layer formula)
            n_src = A.shape[0]
            bytes_ = 4 * (min(n_src, rows) * Cin + rows * Cout + m.K * Cin * Cout) + 4 * rows
            timer.records.append((e0, e1, bytes_, 2.0 * rows * Cin * Cout))
        me._gemm = timed

    def summary(self):
        import ctypes
        from lidog_amd import _lib
        buf = (ctypes.c_double * 4)()
        if _lib.load().lidog_trunk_gemm_timing_read(buf) != 0:
            raise RuntimeError(_lib.load().lidog_last_error().decode())
        self.native = [a + b for a, b in zip(self.native, buf)]
        n = len(self.records) + int(self.native[0])
        if not n:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in self.records) + self.native[1]
        by = sum(r[2] for r in self.records) + self.native[3]
        fl = sum(r[3] for r in self.records) + self.native[2]
        return dict(launches=n, total_ms=ms, bytes=by, flops=fl)


def usable_cpus():
    """(cpus this process may really use, how that was found): the scheduler affinity, cut by the cgroup CPU quota when
    one is set -- a GPU box hands a one-GPU job a share of the host's threads, os.cpu_count() still says the whole host"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = f"affinity {n}"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = float(quota) / period
                how += f", cgroup quota {q:.1f}"
                n = max(1, min(n, int(q + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, how


def _cpu_model(config):
    import oracle.me_cpu as OME
    from oracle.ref_torch import Encoder2DRef, sparse2super_ref
    from lidog_amd.minkunet import make_models
    torch.manual_seed(0)
    cls = make_models(OME, Encoder2DRef, lambda x, bound, voxel, pool: sparse2super_ref(x.C, x.F, bound, voxel, pool))
    model = cls.MinkUNet34BEV(in_channels=1, out_channels=7, D=3, initial_kernel_size=5, decoder_2d_level=["block8"],
                              mapping_bound_2d=50.0)
    model.train()
    return model, torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)


def cpu_worker(config, threads, first_seed, steps):
    """one worker process of the whole-host CPU leg (`python bench.py --cpu-worker ...`, never touches the GPU): the
    oracle's training step on `steps` scans after one untimed step; prints {"steps", "seconds"}"""
    import oracle.me_cpu as OME
    from oracle.ref_torch import soft_dice_loss_ref, dice_loss_ref
    from oracle.me_cpu._lib import lib as _olib
    from lidog_amd import synth
    OME.set_mode("blas")
    torch.set_num_threads(threads)
    _olib().orc_set_threads(threads)
    model, opt = _cpu_model(config)
    scans = [synth.make_batch([first_seed + i], config, device="cpu") for i in range(steps + 1)]

    def one_step(b):
        st = OME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
        sem, bev = model(st, is_train=True)
        loss = 0.5 * soft_dice_loss_ref(sem.F, b["source_sem_labels0"]) + \
            0.5 * dice_loss_ref(bev["block8"].view(-1, 7), b["source_bev_labels0"]["block8"].view(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()
    one_step(scans[0])
    t0 = time.time()
    for b in scans[1:]:
        one_step(b)
    print(json.dumps({"steps": steps, "seconds": time.time() - t0, "t0": t0, "t1": time.time()}), flush=True)


def cpu_whole_host(config, threads_per_worker, single):
    """cpu_baseline.whole_host (north_star: "timed on the same box's host cores"): floor(usable cpus / threads) worker
    PROCESSES x `threads_per_worker` threads, each running the oracle's training step on its own scans, all at once; the
    figure is the sum of the workers' scans/s.  Workers are fresh children (`--cpu-worker`) started by a process that may
    hold the GPU: they never touch it.  With one worker's worth of cpus the single-process figure IS the whole-host one."""
    import subprocess
    ncpu, how = usable_cpus()
    forced = os.environ.get("LIDOG_CPU_BASELINE_WORKERS")
    workers = int(forced) if forced else max(1, ncpu // threads_per_worker)
    workers = min(workers, 32)
    # a worker holds the oracle's activations of one 120 k-point scan plus torch's autograd graph: about 10 GB at the peak
    # of its backward pass; never start more than the memory that is free right now can hold (a host OOM kill would take
    # the parent, and with it the headline line, along)
    mem_cap = None
    try:
        avail_kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
        mem_cap = max(1, int(avail_kb * 1024 * 0.6 // (10 << 30)))
    except (OSError, IndexError, ValueError):
        pass
    if mem_cap is not None and not forced:
        workers = min(workers, mem_cap)
    out = {"usable_cpus": ncpu, "usable_cpus_source": how, "host_cpu_count": os.cpu_count(),
           "threads_per_worker": threads_per_worker, "workers": workers, "workers_memory_cap": mem_cap, "unit": "scans/s"}
    if workers == 1:
        out.update(value=single, cores=threads_per_worker, same_as_single_process=True,
                   sample="one worker's worth of usable cpus: the single-process figure above is the whole-host figure")
        return out
    env = dict(os.environ, OMP_NUM_THREADS=str(threads_per_worker), MKL_NUM_THREADS=str(threads_per_worker),
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", config,
                               str(threads_per_worker), str(100 + 10 * w), "2"],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for w in range(workers)]
    rates, spans = [], []
    # ONE deadline for the whole leg (the workers run at the same time: a worker's three steps take about 15 s alone, a
    # few times that next to the others); whoever has not answered by then is killed, reaped and not counted
    deadline = time.time() + float(os.environ.get("LIDOG_CPU_BASELINE_DEADLINE_S", "180"))
    for p in procs:
        try:
            o, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            d = json.loads([l for l in o.splitlines() if l.startswith("{")][-1])
            rates.append(d["steps"] / d["seconds"])
            spans.append((d["t0"], d["t1"]))
        except Exception:       # a worker that died (memory) or overran is simply not counted, and said so
            p.kill()
            try:
                p.communicate(timeout=10)
            except Exception:
                pass
    overlap = (min(t1 for _, t1 in spans) - max(t0 for t0, _ in spans)) / max(t1 - t0 for t0, t1 in spans) if spans else 0.0
    out.update(value=sum(rates), cores=threads_per_worker * len(rates), workers_finished=len(rates),
               per_worker_scans_per_s=[round(r, 4) for r in rates], timed_windows_overlap=round(overlap, 3),
               sample=f"{len(rates)} of {workers} worker processes x {threads_per_worker} threads, 2 training steps each on own "
                      f"scans after one untimed step, oracle blas mode, started together; value = sum of the workers' rates")
    return out


def cpu_baseline(config, steps_budget_s=14.0, threads=None):
    """The CPU oracle (ME-equivalent restatement, per-offset gather -> BLAS GEMM -> scatter-add) timed on this box's
    host cores on a BOUNDED sample of the same workload: the headline configuration's training step and
    forward pass, one scan per step, and configs[0] (`source8k`, MinkUNet34, SoftDICE only) over 8 scans.  The thread
    count comes from a 3-point sweep of the forward pass recorded in the line.  A stated baseline, never the target."""
    import oracle.me_cpu as OME
    from oracle.ref_torch import Encoder2DRef, sparse2super_ref, soft_dice_loss_ref, dice_loss_ref
    from lidog_amd.minkunet import make_models
    from lidog_amd import synth
    from oracle.me_cpu._lib import lib as _olib
    OME.set_mode("blas")
    prev_threads = torch.get_num_threads()
    ncpu = os.cpu_count() or 1

    def set_threads(n):
        torch.set_num_threads(n)
        _olib().orc_set_threads(n)

    model, opt = _cpu_model(config)
    scans = [synth.make_batch([seed], config, device="cpu") for seed in (0, 1, 2)]

    def forward_only(b):
        with torch.no_grad():
            st = OME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
            model(st, is_train=True)

    def one_step(b):
        st = OME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
        sem, bev = model(st, is_train=True)
        loss = 0.5 * soft_dice_loss_ref(sem.F, b["source_sem_labels0"]) + \
            0.5 * dice_loss_ref(bev["block8"].view(-1, 7), b["source_bev_labels0"]["block8"].view(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()

    # thread count: the layer GEMMs are small, past a few dozen threads the two OpenMP runtimes (torch's and the
    # oracle's) only fight; three points around the usual optimum, one forward pass each after an untimed one
    forced = threads or (int(os.environ["LIDOG_CPU_BASELINE_THREADS"]) if "LIDOG_CPU_BASELINE_THREADS" in os.environ else None)
    sweep = {}
    if forced:
        best = forced
        set_threads(best)
        forward_only(scans[0])
    else:
        points = sorted({max(1, min(ncpu, n)) for n in (16, 32, 64)})
        set_threads(points[0])
        forward_only(scans[0])           # allocations, thread pools: not timed
        for n in points:
            set_threads(n)
            t = time.time()
            forward_only(scans[1])
            sweep[str(n)] = round(time.time() - t, 3)
        best = int(min(sweep, key=sweep.get))
        set_threads(best)
    # training step: one untimed step, then one step per scan on DIFFERENT scans until the budget is used, at least two
    t_warm = time.time()
    one_step(scans[0])
    t_warm = time.time() - t_warm
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 + (time.time() - t0) / max(n, 1) < steps_budget_s and n < 6):
        one_step(scans[(n + 1) % 3])
        n += 1
    dt = time.time() - t0
    # forward pass only (the validation path's arithmetic): two scans
    t1 = time.time()
    for i in range(2):
        forward_only(scans[(i + 2) % 3])
    dt_f = time.time() - t1
    # configs[0]: train_source.py's workload, the reference's own CPU-runnable case -- MinkUNet34 (no BEV head), SoftDICE,
    # 8 k-point scans at 0.1 m; one untimed step, then 8 training steps and 8 forward passes, one scan each
    c1 = None
    try:
        m1 = make_models(OME).MinkUNet34(in_channels=1, out_channels=7, D=3)
        m1.train()
        o1 = torch.optim.Adam(m1.parameters(), lr=1e-2)
        s1 = [synth.make_batch([seed], "source8k", device="cpu") for seed in range(9)]

        def step1(b, train=True):
            st = OME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
            if not train:
                with torch.no_grad():
                    m1(st)
                return
            loss = soft_dice_loss_ref(m1(st).F, b["source_sem_labels0"])
            o1.zero_grad()
            loss.backward()
            o1.step()
        step1(s1[8])
        t2 = time.time()
        for b in s1[:8]:
            step1(b)
        dt1 = time.time() - t2
        t3 = time.time()
        for b in s1[:8]:
            step1(b, train=False)
        dt1f = time.time() - t3
        c1 = {"workload": "configs[0]: MinkUNet34, SoftDICE, 8 synthetic source8k scans (8 000 points, 0.1 m), one per step",
              "train_scans_per_s": 8 / dt1, "forward_scans_per_s": 8 / dt1f, "sample": f"8 steps {dt1:.1f} s, 8 forward passes {dt1f:.1f} s"}
    except Exception as e:      # the headline figure must not depend on the secondary one
        c1 = {"error": repr(e)}
    OME.set_mode("exact")
    torch.set_num_threads(prev_threads)
    try:
        whole = cpu_whole_host(config, best, n / dt)
    except Exception as e:      # the headline line must not depend on this leg
        whole = {"error": repr(e)}
    return {"value": n / dt, "unit": "scans/s", "cores": best, "host_cpu_count": ncpu, "whole_host": whole,
            "torch_num_threads": best, "kind": "port",
            "thread_sweep_forward_s_per_scan": sweep or None,
            "forward_only_scans_per_s": 2 / dt_f,
            "configs0_source8k": c1,
            "sample": f"{n} training steps, one per synthetic {config} scan (seeds 1, 2, 0, ...), after one untimed step "
                      f"({t_warm:.1f} s); MinkUNet34BEV B=50, oracle blas mode (per-offset gather->GEMM->scatter-add), "
                      f"{dt:.1f} s timed; forward only: 2 scans in {dt_f:.1f} s"}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks as fresh children -- the driver's own
    line, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py ...` -- BEFORE this process touches the GPU (a process that has initialised HIP must not be replaced or
    forked), relay what they print (rank 0's JSON line) and exit with their code.  Every rank supervises itself
    (supervise_rank: silence for LIDOG_BENCH_WATCHDOG_S seconds, default 150, ends the rank's worker and starts the
    safe-mode fallback); this process is the outer guard: when NOTHING arrives for that long plus a minute -- not even
    the supervisors' own messages -- the children's process group is killed and the parent exits non-zero, so a hung
    collective can never hang the caller."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.environ.get("LIDOG_BENCH_RANK_SCRIPT", os.path.abspath(__file__))] + sys.argv[1:]   # (script override: tests)
    limit = float(os.environ.get("LIDOG_BENCH_WATCHDOG_S", str(WATCHDOG_DEFAULT_S)))
    limit += min(60.0, limit)        # the ranks' own supervisors act first
    env = dict(os.environ, LIDOG_BENCH_LAUNCHED_BY_PARENT="1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, start_new_session=True, text=True)
    last = [time.time()]
    timed_out = [False]

    def watchdog():
        while proc.poll() is None:
            if time.time() - last[0] > limit:
                timed_out[0] = True
                try:
                    os.killpg(proc.pid, signal.SIGTERM)
                    time.sleep(10)
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                return
            time.sleep(1.0)

    threading.Thread(target=watchdog, daemon=True).start()
    for line in proc.stdout:
        last[0] = time.time()
        # rank 0's result goes to stdout, everything else (launcher banners, warnings) to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        (sys.stdout if line.startswith("{") else sys.stderr).flush()
    rc = proc.wait()
    if timed_out[0]:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run printed nothing for {limit:.0f} s and was killed\n")
        rc = rc or 124
    if rc == 0:
        # phase 2: rank 0's line is out and every rank has torn down and left -- nothing below can change the result
        probe_phase([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                     "--master-addr", "127.0.0.1", "--master-port", str(_free_port())], dict(os.environ))
    sys.exit(rc if rc >= 0 else 128 - rc)


def supervise_rank():
    """What a rank of an N > 1 run IS, whoever launched it (bench.py's own launcher above or the driver's
    `python -m torch.distributed.run ... bench.py --gpus N` line): a supervisor that never touches the GPU and does the
    rank's work in a CHILD (the same command line with LIDOG_BENCH_WORKER=1; its stdout is this process's stdout, its
    stderr is relayed line by line).  Three RCCL communicators, a bucket stream and the executor's side streams have
    never met a second GPU on the build's one-GPU boxes, so the first N > 1 step may hang; then

    * a worker that reports nothing (beat()) for LIDOG_BENCH_WATCHDOG_S seconds (default 150) is killed with its
      process group -- the supervisor holds no GPU state, so nothing of the hung run survives in it --,
    * a FRESH worker runs the same measurement in safe mode (LIDOG_DP_SAFE=1, lidog_amd/__init__.py: one communicator,
      gradient buckets on the compute stream after backward, no peer mailboxes, no side streams) on a rendezvous port
      derived from the first one, without the launcher's TORCHELASTIC_* variables (rank 0's worker serves the store
      itself), and rank 0 prints that line with "mode": "safe-fallback",
    * and the supervisor exits NON-ZERO either way (124: the fallback printed its line; 125: it failed too), so a
      degraded figure can never pass for the real one.

    A worker that exits non-zero gets the same single fallback attempt (the other ranks reach theirs through the silence
    limit; the rendezvous of the fallback group waits for them).  SIGTERM / SIGINT (the launcher tearing the job down) end
    the worker's process group first.
    LIDOG_BENCH_FALLBACK=0 skips the second run; LIDOG_BENCH_WORKER_SCRIPT replaces the worker (tests)."""
    import signal
    import subprocess
    import threading
    limit = float(os.environ.get("LIDOG_BENCH_WATCHDOG_S", str(WATCHDOG_DEFAULT_S)))
    script = os.environ.get("LIDOG_BENCH_WORKER_SCRIPT", os.path.abspath(__file__))
    current = [None]
    reached_result = [False]

    def stop_child():
        p = current[0]
        if p is None or p.poll() is not None:
            return
        for sig, wait in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
            try:
                os.killpg(p.pid, sig)
            except ProcessLookupError:
                return
            try:
                p.wait(timeout=wait)
                return
            except subprocess.TimeoutExpired:
                continue

    def on_signal(signum, frame):
        stop_child()
        os._exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)

    def run_worker(extra, fallback):
        env = dict(os.environ, LIDOG_BENCH_WORKER="1", **extra)
        if fallback:
            env = {k: v for k, v in env.items() if not k.startswith("TORCHELASTIC_") and k != "LIDOG_BENCH_FAULT"}
        p = subprocess.Popen([sys.executable, script] + sys.argv[1:], stderr=subprocess.PIPE, env=env,
                             start_new_session=True, text=True)
        current[0] = p
        last = [time.time()]

        def pump():
            for line in p.stderr:
                last[0] = time.time()
                if "result line out" in line or "timed region done" in line:
                    reached_result[0] = True       # the measurement is out: whatever happens later, no second line
                sys.stderr.write(line)
                sys.stderr.flush()

        t = threading.Thread(target=pump, daemon=True)
        t.start()
        while True:
            rc = p.poll()
            if rc is not None:
                t.join(timeout=5)
                return rc, False
            if time.time() - last[0] > limit:
                stop_child()
                return None, True
            time.sleep(0.25)

    rc, silent = run_worker({}, False)
    if not silent and rc == 0:
        sys.exit(0)
    fallback = os.environ.get("LIDOG_BENCH_FALLBACK", "1") != "0" and not reached_result[0]
    if silent:
        beat(f"worker reported nothing for {limit:.0f} s and was killed")
        if not fallback:
            sys.exit(124)
    else:
        # a worker that FAILED (a communicator that cannot be set up between these GPUs, a launch error): the other ranks'
        # workers are then stuck in a collective and their supervisors arrive here through the silence limit; the safe mode
        # takes none of this library's communicators, so it is worth one attempt.  The exit code stays non-zero either way.
        code = rc if rc >= 0 else 128 - rc
        beat(f"worker exited with code {code}")
        if not fallback:
            sys.exit(code)
    port = 20000 + (int(os.environ.get("MASTER_PORT", "29500")) + 7919) % 40000      # the same on every rank
    beat(f"starting the safe-mode fallback (LIDOG_DP_SAFE=1, rendezvous port {port})")
    rc, silent = run_worker(dict(LIDOG_DP_SAFE="1", LIDOG_BENCH_MODE="safe-fallback", MASTER_ADDR="127.0.0.1",
                                 MASTER_PORT=str(port), LIDOG_BENCH_PROBE="0"), True)
    if silent:
        beat("the safe-mode fallback went silent too and was killed")
    sys.exit(124 if (rc == 0 and not silent) else 125)


PROBE_SCRIPT = os.path.join(REPO, "scripts", "micro_peer_allreduce.py")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def probe_phase(prefix, env):
    """Second, short phase of an N > 1 run, AFTER the headline line has been printed and flushed and the ranks of the
    timed run are gone: fresh processes (`prefix` + scripts/micro_peer_allreduce.py) set up the one-shot peer all-reduce
    in probe mode and time one 193-double statistics message through it, through this library's RCCL communicator and
    through torch.distributed -- the path the SyncBatchNorm design rests on has never run between
    two GPUs, and no training step may be its first run.  Own watchdog (LIDOG_BENCH_PROBE_WATCHDOG_S, at most 120 s: the
    process group is killed), everything it prints goes to stderr, its exit code is ignored.  LIDOG_BENCH_PROBE=0 skips it."""
    import signal
    import subprocess
    if os.environ.get("LIDOG_BENCH_PROBE", "1") == "0":
        return None
    limit = min(120.0, float(os.environ.get("LIDOG_BENCH_PROBE_WATCHDOG_S", "120")))
    beat(f"peer probe phase (at most {limit:.0f} s, result ignored)")
    script = os.environ.get("LIDOG_BENCH_PROBE_SCRIPT", PROBE_SCRIPT)       # (override: tests)
    env = {k: v for k, v in env.items() if k not in ("LIDOG_PEER_ALLREDUCE", "LIDOG_PEER_FAULT")}
    try:
        proc = subprocess.Popen(prefix + [script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env,
                                start_new_session=True, text=True)
    except OSError as e:
        sys.stderr.write(f"bench.py: peer probe not started: {e}\n")
        return None
    try:
        out, _ = proc.communicate(timeout=limit)
        code = proc.returncode
    except subprocess.TimeoutExpired:
        # SIGTERM first: the launcher ends its workers itself; then SIGKILL for whatever is left of the group.  Never wait
        # for ever on the pipe (an orphaned worker could hold it open)
        out = ""
        for sig, wait in ((signal.SIGTERM, 15), (signal.SIGKILL, 5)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                pass
            try:
                out, _ = proc.communicate(timeout=wait)
                break
            except subprocess.TimeoutExpired:
                continue
        code = "killed"
        sys.stderr.write(f"bench.py: peer probe killed after {limit:.0f} s (ignored)\n")
    keep = [l for l in (out or "").splitlines() if "peer_probe" in l or "Error" in l or "error" in l]
    sys.stderr.write("\n".join(keep[-20:]) + ("\n" if keep else ""))
    sys.stderr.write(f"bench.py: peer probe finished ({code}); its exit code is ignored\n")
    sys.stderr.flush()
    return code


# LIDOG_* variables that only say HOW the bench is launched (ranks, watchdog, which legs run): they do not change what
# a step computes or which kernels it takes.  Everything else that is set is an experiment switch and is echoed in
# `config.env`, so that a line can never silently describe another workload or another build of the step.
_LAUNCH_ONLY_ENV = ("LIDOG_BENCH_LAUNCHED_BY_PARENT", "LIDOG_BENCH_WATCHDOG_S", "LIDOG_BENCH_RANK_SCRIPT",
                    "LIDOG_BENCH_PROBE", "LIDOG_BENCH_PROBE_WATCHDOG_S", "LIDOG_CPU_BASELINE_THREADS",
                    "LIDOG_CPU_BASELINE_WORKERS", "LIDOG_CPU_BASELINE_DEADLINE_S", "LIDOG_BENCH_WORKER",
                    "LIDOG_BENCH_WORKER_SCRIPT", "LIDOG_BENCH_FALLBACK", "LIDOG_BENCH_MODE", "LIDOG_BENCH_PROBE_SCRIPT")


def experiment_env():
    """every LIDOG_* variable set in this process's environment that can change what the step runs"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("LIDOG_") and k not in _LAUNCH_ONLY_ENV}


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        # checked before any collective or GPU work: every rank sees the same mismatch and leaves with the same message
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; run "
                 f"`python bench.py --gpus {args.gpus}` (it starts its ranks itself) or launch {args.gpus} ranks")
    if world > 1 and os.environ.get("LIDOG_BENCH_WORKER") != "1":
        return supervise_rank()
    beat("worker up" + (f" (mode {os.environ['LIDOG_BENCH_MODE']})" if "LIDOG_BENCH_MODE" in os.environ else ""))
    # stdout carries ONE line, rank 0's JSON.  RCCL prints a version banner to file descriptor 1 when its first
    # communicator comes up, and other native libraries may do the same: in a distributed run descriptor 1 is pointed at
    # stderr for the life of the process and the result line is written to the saved descriptor (emit() below).
    result_fd = None
    if world > 1 or os.environ.get("LIDOG_BENCH_SINGLE_RANK_DP") == "1":
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)

    def emit(line):
        if result_fd is None:
            print(line, flush=True)
        else:
            os.write(result_fd, (line + "\n").encode())
    # test hooks only: LIDOG_BENCH_ONE_GPU=1 runs every rank on cuda:0 over gloo so the multi-rank control flow
    # (SyncBN conversion, gradient buckets, barriers, max-over-ranks) can be exercised on a 1-GPU box
    one_gpu = os.environ.get("LIDOG_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    # LIDOG_BENCH_SINGLE_RANK_DP=1: a ONE-rank RCCL process group with every data-parallel code path switched on
    # (SyncBN statistics all-reduces, gradient buckets, second backward stream): the collectives of the N > 1
    # runs exercised against the real RCCL library on a 1-GPU box
    single_dp = world == 1 and os.environ.get("LIDOG_BENCH_SINGLE_RANK_DP") == "1"
    if world > 1 or single_dp:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:   # the launcher's rendezvous port is the only source for N > 1
                assert single_dp, "MASTER_PORT must come from the launcher (torch.distributed.run)"
                import socket
                with socket.socket() as sk:       # one-rank group on a 1-GPU box: any free port
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        beat(f"process group up ({dist.get_backend()}, {world} ranks)")

    import lidog_amd
    import lidog_amd.me as ME
    from lidog_amd import synth
    from lidog_amd.train import build_model, build_step

    if single_dp:
        from lidog_amd.trainer import GradientBuckets
        ME.MinkowskiSyncBatchNorm.single_rank = GradientBuckets.single_rank = True
    fault = os.environ.get("LIDOG_BENCH_FAULT", "").split(":")
    if fault[0] == "bucket_hang" and int(fault[1]) == rank:
        # test hook of the watchdog (tests/test_gpu_dist.py): this rank never issues its first gradient bucket, so every
        # other rank waits in that all-reduce for ever -- what a collective that hangs between two GPUs looks like
        from lidog_amd.optim import GradientBuckets as _GB

        def _never(self, b):
            beat("injected fault: hanging in front of a gradient bucket (LIDOG_BENCH_FAULT)")
            time.sleep(1e6)
        _GB._reduce = _never
    timer = GemmTimer()
    head_timer = None
    if not args.no_kernel_timing:
        timer.wrap(ME)
        from lidog_amd import bev as _bev
        head_timer = _bev.HEAD_TIMER = _bev.HeadTimer()

    # model, SyncBN + data-parallel wiring, optimiser and step object come from the owning driver (lidog_amd/train.py,
    # the restatement of train_lidog.py:42-75,227-231 and trainer_lighting_2d.py:349-360); this file only times steps
    torch.manual_seed(1234)  # pipeline.seed of configs/lidog/single/semantickitti.yaml
    model, step, _ = build_step(build_model("MinkUNet34BEV", bound_2d=50.0), "MinkUNet34BEV", optimizer="Adam", lr=1e-3,
                                weight_decay=1e-4, source_weights=(0.5, 0.5))

    # two distinct batches per rank, resident in HBM, cycled (scan seeds differ per rank)
    base = rank * 2 * args.batch
    batches = [synth.make_batch(range(base + i * args.batch, base + (i + 1) * args.batch), args.config, "cuda")
               for i in range(2)]
    n_vox = sum(b["coords_int"].shape[0] for b in batches) / (2 * args.batch)
    beat("model, optimiser and batches ready")
    # the workload is BASELINE.md's: scan seed 0 must have SURVEY.md 8(d)'s per-stride voxel counts
    seed0_counts = None
    if args.config in synth.BASELINE_COUNTS:
        seed0_counts = list(synth.stride_counts(synth.scan_voxels(0, args.config)[0]))
        assert tuple(seed0_counts) == synth.BASELINE_COUNTS[args.config], (seed0_counts, synth.BASELINE_COUNTS[args.config])

    def sync():
        if world > 1 or single_dp:
            dist.barrier()
        torch.cuda.synchronize()

    # The coordinate maps of batch i+1 are built on a side stream while step i still runs (every timed step
    # builds one set of maps: the last one prefetches for a step that never comes, the first one was prefetched
    # by the warm-up).  --no-prefetch builds them inside the forward pass instead.
    ready = torch.cuda.Event()
    ready.record()
    torch.cuda.synchronize()

    # Data-parallel runs put the step on a high-priority stream: the dependent chain (with its SyncBN collectives)
    # is served first, the weight-gradient and coordinate-map streams (default priority) fill what is left
    # (measured on one rank with every collective active: 74.7 -> 75.5 scans/s).  LIDOG_MAIN_STREAM_PRIORITY
    # overrides (-1 = high, "none" = run on the default stream).
    prio = os.environ.get("LIDOG_MAIN_STREAM_PRIORITY", "-1" if (world > 1 or single_dp) else "none")
    main_stream = torch.cuda.Stream(priority=int(prio)) if prio != "none" else None

    def run(i):
        if main_stream is not None:
            with torch.cuda.stream(main_stream):
                return run_(i)
        return run_(i)

    def run_(i):
        if args.no_prefetch:
            return step.training_step(batches[i % 2])
        return step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)

    for i in range(args.warmup):
        out = run(i)
        if i == 0:
            sync()      # the first step (every collective's first use) has completed on every rank
            beat("first step done")
    sync()
    beat(f"{args.warmup} warm-up steps done")
    # The timed region is EXACTLY --steps steps between two (barrier + synchronize) brackets, max over ranks.  A block
    # of 20 steps lasts one second, inside a process whose life is dominated by start-up and the CPU-baseline leg; so
    # the block is repeated until about --min-seconds of step time have been measured (same count on every rank: it
    # follows from the all-reduced time of the first block) and the MEDIAN block is the one reported -- every block's
    # figure is in "blocks_ms_per_step".
    timed_steps = 0
    blocks = []
    step_no = args.warmup
    n_blocks = 1
    while len(blocks) < n_blocks:
        t0 = time.perf_counter()
        for i in range(args.steps):
            # HIP events around the dominant kernel's launches on every 10th timed step (100 launches each): an event
            # pair costs a few microseconds of queue time per launch, 0.7 ms per step when every step is instrumented (every
            # 4th step, rounds 1-4, put 0.17 ms of that into the reported step time)
            timer.enabled = not args.no_kernel_timing and i % 10 == 0
            if head_timer is not None:
                head_timer.enabled = timer.enabled
            timed_steps += int(timer.enabled)
            out = run(step_no)
            step_no += 1
        sync()
        dt_block = time.perf_counter() - t0
        timer.enabled = False
        if head_timer is not None:
            head_timer.enabled = False
        if world > 1:
            t = torch.tensor([dt_block], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_block = float(t.item())
        blocks.append(dt_block)
        beat(f"block {len(blocks)}: {1e3 * dt_block / args.steps:.2f} ms per step")
        if len(blocks) == 1:
            n_blocks = max(1, min(args.max_blocks, int(-(-args.min_seconds // dt_block))))
    ordered = sorted(blocks)
    dt = ordered[len(ordered) // 2]           # median block (the slower of the middle two for an even count)
    total_timed_steps = args.steps * len(blocks)
    loss = float(out["loss"])
    # data-parallel invariant: after any number of steps every rank holds the same parameters and the same BatchNorm
    # running statistics (identical start, identical all-reduced gradients and statistics).  Checked bit for bit over
    # the ranks -- a collective that misbehaved (lost bucket, stale statistics message) shows up here.
    replicas_identical = None
    peer_error = None
    if world > 1 or single_dp:
        flat = step.opt.flat.flat
        # the two BatchNorm2d of Encoder2D keep per-rank running statistics (they are not SyncBatchNorm in the reference
        # either, train_lidog.py:228 converts the Minkowski ones only): not part of the invariant
        stats = torch.cat([b.detach().float().flatten() for n, b in model.named_buffers()
                           if "running" in n and not n.startswith("encoders2d")])

        def same_as_rank0(t):
            """every element bit-identical to rank 0's copy, on every rank (rank 0's tensor travels to all ranks and is
            compared there with torch.equal on the raw bits; the verdicts are MIN-reduced)"""
            ref = t.detach().clone()
            dist.broadcast(ref, src=0)
            same = torch.equal(ref.view(torch.int32), t.detach().view(torch.int32))
            v = torch.tensor([1 if same else 0], device=t.device, dtype=torch.int32)
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            return bool(v.item())
        replicas_identical = {"parameters": same_as_rank0(flat), "syncbn_running_statistics": same_as_rank0(stats)}
        # the peer all-reduce's error word, agreed over the ranks (collective: every rank calls it)
        from lidog_amd.comm import transport
        try:
            transport().check()
        except RuntimeError as e:
            peer_error = str(e)
    # latency of the statistics all-reduce (193 doubles = one 96-channel message), back to back on one stream, through
    # every transport this run has: what the 241 messages per step cost on the dependent chain
    collective_us = None
    if world > 1 or single_dp:
        from lidog_amd import _lib as _L
        from lidog_amd.comm import transport
        tr = transport()
        msg = torch.ones(193, dtype=torch.float64, device="cuda")
        ways = {}
        pc = tr.peer or tr.peer_probe   # set up and self-tested on every rank; used by the step only with LIDOG_PEER_ALLREDUCE=1
        if pc and peer_error is None:
            def _peer():
                tr.peer_bind(pc)
                _L.call("lidog_peer_allreduce_f64", pc, _L.ptr(msg), msg.numel())
            ways["peer_one_shot"] = _peer
        if tr.comm_bn:
            ways["rccl"] = lambda: _L.call("lidog_allreduce_f64", _L.ptr(msg), msg.numel(), tr.comm_bn)
        ways["torch_distributed"] = lambda: dist.all_reduce(msg)
        collective_us = {}
        for name, fn in ways.items():
            reps = 200 if (name != "torch_distributed" or dist.get_backend() == "nccl") else 5   # gloo stages through the host
            for _ in range(3):
                fn()
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                msg.fill_(1.0)
                fn()
            e1.record()
            torch.cuda.synchronize()
            collective_us[name] = round(1e3 * e0.elapsed_time(e1) / reps, 2)
        sync()
    eval_rate = None
    if world == 1:
        # secondary figure: forward-only validation path, is_train=False, same batches
        from lidog_amd.evaluate import Predictor
        run_eval = Predictor(model)
        for i in range(3):   # warm-up: the first call records the map trace, the others prefetch from it
            run_eval(batches[i % 2]["coords_int"], batches[i % 2]["source_features0"], batches[(i + 1) % 2]["coords_int"])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_eval = 24
        for i in range(3, 3 + n_eval):
            run_eval(batches[i % 2]["coords_int"], batches[i % 2]["source_features0"], batches[(i + 1) % 2]["coords_int"])
        torch.cuda.synchronize()
        eval_rate = n_eval * args.batch / (time.perf_counter() - t1)
    if rank == 0:
        value = world * args.batch * args.steps / dt
        res = {"metric": "LiDAR scans/sec, MinkUNet34+BEV training step @120k pts", "value": value, "unit": "scans/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{n_vox:.0f} voxels/scan ({args.config} synthetic, {synth.CONFIGS[args.config]['voxel']} m), "
                                      f"MinkUNet34+BEV B=50 train step"
                                      f"{' (configs[1])' if (args.config, args.batch) == ('kitti120k', 4) else ''}, "
                                      f"bs={args.batch}/GPU, SoftDICE+DICE, Adam",
                          "voxels_per_scan": n_vox, "seed0_stride_counts": seed0_counts,
                          "global_batch": world * args.batch,
                          "parallelism": f"dp{world}" + ("+syncbn" if world > 1 or single_dp else ""),
                          "env": experiment_env(),
                          # how the weight-gradient work items were cut (me._wgrad_chunk: 1 / 2 = whole rounds of the
                          # kernel's resident workgroups, 0 = not; the default follows the stream mode, so one-stream and
                          # two-stream runs differ in the last bits of the weight gradients unless LIDOG_WGRAD_FIT pins it)
                          "wgrad_fit": ME._WGRAD_FIT if ME._WGRAD_FIT >= 0 else (0 if ME._WgradLane.enabled else 1)},
               "blocks_ms_per_step": [round(1e3 * b / args.steps, 3) for b in blocks], "loss": loss,
               # "safe-fallback": the line of supervise_rank's second run after the first one went silent
               "mode": os.environ.get("LIDOG_BENCH_MODE", "normal")}
        if os.environ.get("LIDOG_DP_SAFE") == "1":
            res["config"]["dp_safe"] = "LIDOG_DP_SAFE=1: torch.distributed collectives only, buckets after backward, no side streams"
        if world > 1 or single_dp:
            from lidog_amd.comm import transport
            from lidog_amd import _lib as _L
            tr = transport()
            res["config"]["collectives"] = tr.kind     # native = this library's RCCL communicators
            res["config"]["statistics_allreduce"] = "peer one-shot" if tr.peer else f"{tr.kind} (RCCL)" if tr.kind == "native" \
                else f"torch.distributed ({dist.get_backend()})"
            res["config"]["peer_note"] = tr.peer_note
            # how many ranks the communicators really span, as RCCL itself counts them (not WORLD_SIZE echoed back)
            res["rccl_ranks_seen"] = {"statistics": _L.load().lidog_comm_count(tr.comm_bn) if tr.comm_bn else None,
                                      "gradient_buckets": _L.load().lidog_comm_count(tr.comm_grad) if tr.comm_grad else None,
                                      "torch_process_group": dist.get_world_size(), "backend": dist.get_backend()}
            res["devices_seen"] = torch.cuda.device_count()
            if peer_error:
                res["peer_error"] = peer_error
            res["config"]["trunk_path"] = "executor" if getattr(step, "last_path", "") == "_TrunkFnBackward" else \
                getattr(step, "last_path", "unknown")
            res["replicas_identical"] = replicas_identical   # parameters + running statistics equal on every rank
            res["statistics_allreduce_us"] = collective_us    # per call, 193 doubles, incl. a fill kernel (rank 0's clock)
        if eval_rate is not None:
            res["forward_only_scans_per_s"] = eval_rate
        s = timer.summary()
        if s:
            gbs = s["bytes"] / (s["total_ms"] * 1e-3) / 1e9
            tfl = s["flops"] / (s["total_ms"] * 1e-3) / 1e12
            # HBM bytes per launch from the committed rocprofv3 PMC passes over this same command
            # (scripts/pmc_traffic.py, FETCH_SIZE doubled per MI355X_MICROARCH.md); None if not collected
            traffic = None
            pmc = os.path.join(REPO, "profiles", PMC_TRAFFIC_FILE)
            if os.path.exists(pmc) and args.config == "kitti120k" and args.batch == 4:
                traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
            # The dominant kernel is an exact-f32 MFMA GEMM: its binding roofline is the fp32 matrix rate
            # (157.3 TF/s, equal to the vector fp32 peak).  The HBM view the north_star asks for is reported
            # next to it (algorithmic bytes / launch time against 8 TB/s, and the PMC-measured traffic).
            res["roofline"] = {"bound": "mfma", "achieved": tfl, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tfl / FP32_PEAK_TFLOPS, "traffic": traffic,
                               "traffic_source": f"profiles/{PMC_TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                 f"over the {PMC_TRAFFIC_TAG} build (scripts/profile_round.sh), "
                                                 "not collected by this run" if traffic is not None else None,
                               "kernel": "k_sconv_gemm_mfma / k_sconv_gemm_mfma_ms (gathered GEMM of the sparse convolutions, f32 MFMA)",
                               "note": "launch durations IN the step: since round 5 the forward launches share the chip with "
                                       "the downsample branches on the second stream and the backward ones with the weight "
                                       "gradients (DESIGN.md 3a); the in-kernel clock is 2.2-2.4 GHz (profiles/r06_clock_stamps.txt), "
                                       "so the roof at the held clock is 144-157 TFLOP/s; one-stream durations: "
                                       f"profiles/{PMC_TRAFFIC_TAG}_kernel_stats_train_bs4_one_stream.csv",
                               "algorithmic_flops_per_launch": s["flops"] / s["launches"],
                               "algorithmic_bytes_per_launch": s["bytes"] / s["launches"],
                               "avg_launch_us": 1e3 * s["total_ms"] / s["launches"], "launches": s["launches"],
                               "share_of_step": s["total_ms"] / max(timed_steps, 1) / (1e3 * sum(blocks) / total_timed_steps),
                               "instrumented_steps": timed_steps,
                               "hbm_gbs": gbs, "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": gbs / HBM_PEAK_GBS}
            # SURVEY.md 8(d)'s whole-step view of the headline configuration (scripts/count_work.py:
            # 300.0 GFLOP and 4.38 GB compulsory per scan forward, a training step = 3 x forward): dense-equivalent
            # FLOPs and algorithmic bytes of ONE rank's step over its step time, against the fp32 and HBM roofs
            if args.config == "kitti120k":
                step_s = dt / args.steps
                gflop_step, gb_step = 900.0 * args.batch, 13.2 * args.batch
                res["roofline"]["step_dense_equivalent_gflop"] = gflop_step
                res["roofline"]["step_algorithmic_gb"] = gb_step
                res["roofline"]["step_fp32_frac"] = gflop_step / step_s / 1e3 / FP32_PEAK_TFLOPS
                res["roofline"]["step_hbm_frac"] = gb_step / step_s / HBM_PEAK_GBS
                res["roofline"]["binding_roof"] = "fp32 matrix/vector rate (5.7 ms per scan) rather than HBM (1.65 ms per scan)"
            # 2-D head: the 3x3 stride-2 convolutions of Encoder2D, events on the streams they run on, EXECUTED FLOPs
            # (the support-restricted first convolution counts its padded channel lists, not the dense 147 GFLOP per scan)
            hs = head_timer.summary() if head_timer is not None else None
            if hs and hs["total_ms"] > 0:
                h_tfl = hs["executed_flops"] / (hs["total_ms"] * 1e-3) / 1e12
                res["roofline"]["bev_mfma_frac"] = h_tfl / FP32_PEAK_TFLOPS
                res["roofline"]["bev_head"] = {"kernels": "k_conv_s2 / k_conv_wgrad (dense), k_conv_fwd_act / k_conv_dgrad_act / "
                                                          "k_conv_wgrad_act (support-restricted), incl. their weight repacks",
                                               "achieved_tflops": h_tfl, "executed_gflop_per_step": hs["executed_flops"] / max(timed_steps, 1) / 1e9,
                                               "ms_per_step": hs["total_ms"] / max(timed_steps, 1), "launches": hs["launches"]}
            # PMC traffic of the other two HBM-heavy families of the sparse stack, same passes, same build tag
            import ctypes
            from lidog_amd import _lib as _L2
            wk = (ctypes.c_double * 6)()
            _L2.load().lidog_trunk_work_read(wk)   # algorithmic bytes of the executor's weight-gradient / reduction launches
            alg = {"k_sconv_gemm_mfma": s["bytes"] / s["launches"],
                   "k_sconv_wgrad_mfma": wk[1] / wk[0] if wk[0] else None,
                   "k_sconv_reduce_rows4": wk[3] / wk[2] if wk[2] else None,
                   "k_sconv_os_mfma": wk[5] / wk[4] if wk[4] else None}
            ratios = {}
            for fam, fn in (("k_sconv_gemm_mfma", PMC_TRAFFIC_FILE),
                            ("k_sconv_wgrad_mfma", f"{PMC_TRAFFIC_TAG}_pmc_traffic_sconv_wgrad_mfma.json"),
                            ("k_sconv_reduce_rows4", f"{PMC_TRAFFIC_TAG}_pmc_traffic_sconv_reduce_rows4.json"),
                            ("k_sconv_os_mfma", f"{PMC_TRAFFIC_TAG}_pmc_traffic_sconv_os_mfma.json")):
                f = os.path.join(REPO, "profiles", fn)
                if os.path.exists(f) and args.config == "kitti120k" and args.batch == 4 and alg[fam]:
                    d = json.load(open(f))
                    if d.get("traffic_bytes_per_launch"):
                        ratios[fam] = {"pmc_bytes_per_launch": d["traffic_bytes_per_launch"],
                                       "algorithmic_bytes_per_launch": alg[fam],
                                       "ratio": round(d["traffic_bytes_per_launch"] / alg[fam], 3)}
            if ratios:
                res["roofline"]["traffic_ratio"] = ratios     # PMC HBM bytes / algorithmic bytes per launch
                res["roofline"]["traffic_ratio_source"] = f"profiles/{PMC_TRAFFIC_TAG}_pmc_traffic_*.json"
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.config)
            res["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
            wh = res["cpu_baseline"].get("whole_host") or {}
            if wh.get("value"):
                res["gpu_over_cpu_whole_host"] = value / wh["value"]
        emit(json.dumps(res))
    beat("result line out" if rank == 0 else "timed region done")
    probe_port = None
    if world > 1 and os.environ.get("LIDOG_BENCH_LAUNCHED_BY_PARENT") != "1" and not peer_error:
        # ranks started by somebody else's launcher (the driver's torch.distributed.run line): there is no parent of ours
        # to run the probe phase, so every rank starts its own probe child once it has torn down; the rendezvous port of
        # that second group is agreed while the first group still exists
        box = [_free_port() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        probe_port = box[0]
    if world > 1 or single_dp:
        # orderly teardown: every rank has finished its collectives (barrier + synchronize), then this library's
        # communicators and mailboxes go (ncclCommDestroy, hipIpcCloseMemHandle), then torch's group -- no RCCL object is
        # left to the destructors at interpreter exit
        dist.barrier()
        torch.cuda.synchronize()
        from lidog_amd import comm as _comm
        _comm.reset()
        dist.destroy_process_group()
    if peer_error:      # every rank holds the same verdict (Transport.check is collective): all leave non-zero
        sys.exit(f"bench.py: {peer_error}")
    if probe_port is not None:
        # the headline line is out (printed and flushed above) and this rank holds no communicator any more; the child is a
        # fresh process on this rank's GPU, bounded by its own watchdog, exit code ignored
        # (without the launcher's TORCHELASTIC_* variables: with TORCHELASTIC_USE_AGENT_STORE set, rank 0 would look for the
        # launcher agent's store on the new port instead of serving one)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(probe_port))
        probe_phase([sys.executable], env)


if __name__ == "__main__":
    main()
