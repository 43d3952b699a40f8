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
