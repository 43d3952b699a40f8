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
