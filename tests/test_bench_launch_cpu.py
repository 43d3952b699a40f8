"""`python bench.py --gpus N` with no launcher environment starts its N ranks itself (the driver's own
torch.distributed.run line), relays rank 0's JSON line and leaves with the ranks' exit code; a run that goes silent is
killed by the watchdog.  Exercised here with stand-in rank scripts (LIDOG_BENCH_RANK_SCRIPT): no GPU is touched, the
launch path, the relay, the exit code and the watchdog are."""
import json
import os
import subprocess
import sys
import textwrap
import time

from helpers import REPO


def _run(tmp_path, body, extra_env=None, args=("--gpus", "2", "--steps", "2", "--warmup", "1"), timeout=180):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LIDOG_BENCH_RANK_SCRIPT=str(script), **(extra_env or {}))
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=str(tmp_path))
    return p, time.time() - t0


def test_self_launch_relays_rank0_line_and_exit_code(tmp_path):
    p, _ = _run(tmp_path, """
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")     # MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE from the launcher
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and sys.argv[1:] == ["--gpus", "2", "--steps", "2", "--warmup", "1"]
        dist.barrier()
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "launched_by_parent": os.environ.get("LIDOG_BENCH_LAUNCHED_BY_PARENT")}), flush=True)
        dist.destroy_process_group()
        """)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "launched_by_parent": "1"}, p.stdout


def test_self_launch_fails_when_a_rank_fails(tmp_path):
    p, dt = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)      # rank 0 would wait for ever: the launcher ends it when rank 1 fails
        """)
    assert p.returncode != 0 and dt < 120
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_self_launch_watchdog_kills_a_silent_run(tmp_path):
    p, dt = _run(tmp_path, """
        import time
        time.sleep(600)
        """, extra_env={"LIDOG_BENCH_WATCHDOG_S": "5"})
    assert p.returncode != 0 and dt < 90, (p.returncode, dt)
    assert "was killed" in p.stderr


def test_rank_count_mismatch_is_refused_before_any_gpu_work(tmp_path):
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=180)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_too_few_hardware_queues_are_reported_at_import():
    """lidog_amd sets GPU_MAX_HW_QUEUES=16 when nobody has (it only counts before the HIP runtime starts) and warns when
    the environment already pins fewer queues than a step's streams need"""
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    code = "import os, lidog_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    p = subprocess.run([sys.executable, "-W", "always", "-c", code], env=env, capture_output=True, text=True, cwd=REPO)
    assert p.returncode == 0 and p.stdout.strip() == "16" and "hardware queues" not in p.stderr, p.stderr[-500:]
    p = subprocess.run([sys.executable, "-W", "always", "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="8"),
                       capture_output=True, text=True, cwd=REPO)
    assert p.returncode == 0 and p.stdout.strip() == "8" and "8 hardware queues" in p.stderr, p.stderr[-500:]


# ---------------------------------------------------------------- phase 2: the peer probe behind the headline line
_RANK_OK = """
    import json, os
    import torch.distributed as dist
    dist.init_process_group("gloo")
    dist.barrier()
    if dist.get_rank() == 0:
        print(json.dumps({"n_gpus": dist.get_world_size()}), flush=True)
    dist.destroy_process_group()
    """


def _probe_script(tmp_path, body):
    probe = tmp_path / "probe.py"
    probe.write_text(textwrap.dedent(body))
    return str(probe)


def test_probe_phase_runs_after_the_line_and_its_exit_code_is_ignored(tmp_path):
    """after rank 0's line a SECOND group of fresh processes runs the probe script under the same launcher line; what it
    prints goes to stderr, a failing probe does not change the exit code"""
    probe = _probe_script(tmp_path, """
        import os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        assert "LIDOG_PEER_ALLREDUCE" not in os.environ
        dist.barrier()
        if dist.get_rank() == 0:
            sys.stderr.write('{"peer_probe": {"ranks": %d}}\\n' % dist.get_world_size())
        dist.destroy_process_group()
        sys.exit(7)
        """)
    p, _ = _run(tmp_path, _RANK_OK, extra_env={"LIDOG_BENCH_PROBE_SCRIPT": probe, "LIDOG_PEER_ALLREDUCE": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")] == [{"n_gpus": 2}]
    assert '{"peer_probe": {"ranks": 2}}' in p.stderr and "its exit code is ignored" in p.stderr
    assert "peer_probe" not in p.stdout


def test_probe_phase_is_killed_by_its_own_watchdog(tmp_path):
    probe = _probe_script(tmp_path, """
        import time
        time.sleep(600)
        """)
    p, dt = _run(tmp_path, _RANK_OK, extra_env={"LIDOG_BENCH_PROBE_SCRIPT": probe, "LIDOG_BENCH_PROBE_WATCHDOG_S": "4"})
    assert p.returncode == 0 and dt < 120, (p.returncode, dt)
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1
    assert "peer probe killed" in p.stderr


def test_no_probe_after_a_failed_run_or_when_switched_off(tmp_path):
    probe = _probe_script(tmp_path, """
        import sys
        sys.stderr.write("PROBE RAN\\n")
        """)
    p, _ = _run(tmp_path, """
        import sys
        sys.exit(3)
        """, extra_env={"LIDOG_BENCH_PROBE_SCRIPT": probe})
    assert p.returncode != 0 and "PROBE RAN" not in p.stderr
    p, _ = _run(tmp_path, _RANK_OK, extra_env={"LIDOG_BENCH_PROBE_SCRIPT": probe, "LIDOG_BENCH_PROBE": "0"})
    assert p.returncode == 0 and "PROBE RAN" not in p.stderr


# ---------------------------------------------------------------- every rank of an N > 1 run supervises its own worker
def _supervise(tmp_path, body, extra_env=None, rank=0, timeout=120, master_port="29511"):
    """`bench.py --gpus 2` as ONE rank of a foreign launcher's job (RANK / WORLD_SIZE set by hand, no rendezvous needed:
    the supervisor itself never joins a process group); the worker is a stand-in script"""
    worker = tmp_path / f"worker{rank}.py"
    worker.write_text(textwrap.dedent(body))
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=master_port, TORCHELASTIC_RUN_ID="x", LIDOG_BENCH_WORKER_SCRIPT=str(worker), **(extra_env or {}))
    env.pop("LIDOG_BENCH_WORKER", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=str(tmp_path))
    return p, time.time() - t0


_WORKER = """
    import json, os, sys, time
    assert os.environ["LIDOG_BENCH_WORKER"] == "1" and sys.argv[1:] == ["--gpus", "2", "--steps", "2", "--warmup", "1"]
    safe = os.environ.get("LIDOG_DP_SAFE") == "1"
    sys.stderr.write("worker up\\n"); sys.stderr.flush()
    if not safe:
        %s
    print(json.dumps({"mode": os.environ.get("LIDOG_BENCH_MODE", "normal"), "port": os.environ["MASTER_PORT"],
                      "elastic": [k for k in os.environ if k.startswith("TORCHELASTIC_")],
                      "fault": os.environ.get("LIDOG_BENCH_FAULT"), "probe": os.environ.get("LIDOG_BENCH_PROBE")}), flush=True)
    """


def test_supervised_worker_that_finishes_is_relayed_unchanged(tmp_path):
    p, _ = _supervise(tmp_path, _WORKER % "pass")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["mode"] == "normal" and line["port"] == "29511" and line["elastic"] == ["TORCHELASTIC_RUN_ID"]
    assert "worker up" in p.stderr


def test_silent_worker_is_killed_and_a_safe_mode_worker_prints_the_fallback_line(tmp_path):
    """a hung first N > 1 run must still produce a line -- marked as the fallback, with a non-zero exit"""
    p, dt = _supervise(tmp_path, _WORKER % "time.sleep(600)",
                       extra_env={"LIDOG_BENCH_WATCHDOG_S": "3", "LIDOG_BENCH_FAULT": "bucket_hang:1"})
    assert p.returncode == 124 and dt < 60, (p.returncode, dt, p.stderr[-2000:])
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    # fresh rendezvous port (the same on every rank: a function of the launcher's), no launcher agent store, no fault hook
    assert line["mode"] == "safe-fallback" and line["port"] == str(20000 + (29511 + 7919) % 40000)
    assert line["elastic"] == [] and line["fault"] is None and line["probe"] == "0"
    assert "reported nothing for 3 s and was killed" in p.stderr and "safe-mode fallback" in p.stderr


def test_both_ranks_derive_the_same_fallback_port(tmp_path):
    ports = []
    for rank in (0, 1):
        p, _ = _supervise(tmp_path, _WORKER % "time.sleep(600)", rank=rank, extra_env={"LIDOG_BENCH_WATCHDOG_S": "2"})
        ports.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])["port"])
    assert ports[0] == ports[1] != "29511"


def test_a_fallback_that_hangs_too_ends_with_125_and_no_line(tmp_path):
    p, dt = _supervise(tmp_path, """
        import time
        time.sleep(600)
        """, extra_env={"LIDOG_BENCH_WATCHDOG_S": "2"})
    assert p.returncode == 125 and dt < 60, (p.returncode, dt)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "went silent too" in p.stderr


def test_a_worker_that_fails_gets_one_fallback_attempt_and_the_exit_code_stays_non_zero(tmp_path):
    """a communicator that cannot be set up on the first multi-GPU box must not cost the line either"""
    p, dt = _supervise(tmp_path, _WORKER % "sys.stderr.write('boom\\n'); sys.exit(7)", extra_env={"LIDOG_BENCH_WATCHDOG_S": "30"})
    assert p.returncode == 124 and "boom" in p.stderr and "worker exited with code 7" in p.stderr and dt < 60
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["mode"] == "safe-fallback"
    # without the fallback the worker's code is passed on
    p, _ = _supervise(tmp_path, _WORKER % "sys.exit(7)", extra_env={"LIDOG_BENCH_FALLBACK": "0"})
    assert p.returncode == 7 and not p.stdout.strip()
    # a worker that fails AFTER its line is out (bench.py's peer_error exit) keeps that line the only one
    p, _ = _supervise(tmp_path, """
        import sys
        print('{"peer_error": "x"}', flush=True)
        sys.stderr.write("bench.py[rank 0 +1.0s]: result line out\\n"); sys.stderr.flush()
        sys.exit(1)
        """)
    assert p.returncode == 1 and [l for l in p.stdout.splitlines() if l.startswith("{")] == ['{"peer_error": "x"}']


def test_fallback_can_be_switched_off(tmp_path):
    p, _ = _supervise(tmp_path, _WORKER % "time.sleep(600)",
                      extra_env={"LIDOG_BENCH_WATCHDOG_S": "2", "LIDOG_BENCH_FALLBACK": "0"})
    assert p.returncode == 124 and not p.stdout.strip()


def test_sigterm_to_the_supervisor_ends_the_worker(tmp_path):
    """the launcher tears a job down with SIGTERM: the worker (a process group of its own) must not outlive its supervisor"""
    import signal
    worker = tmp_path / "w.py"
    pidfile = tmp_path / "pid"
    worker.write_text(textwrap.dedent(f"""
        import os, time
        open({str(pidfile)!r}, "w").write(str(os.getpid()))
        time.sleep(600)
        """))
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_PORT="29512", LIDOG_BENCH_WORKER_SCRIPT=str(worker))
    env.pop("LIDOG_BENCH_WORKER", None)
    sup = subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for _ in range(600):
        if pidfile.exists() and pidfile.read_text():
            break
        time.sleep(0.1)
    pid = int(pidfile.read_text())
    sup.send_signal(signal.SIGTERM)
    assert sup.wait(timeout=30) == 128 + signal.SIGTERM
    for _ in range(100):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        os.kill(pid, signal.SIGKILL)
        raise AssertionError("the worker outlived its supervisor")
