"""Soak check of the executor's side streams (csrc/trunk.hip: downsample branches on the second / a third stream): N training
steps of the bench workload with the side streams on and off in two child processes; every parameter and running statistic
must be bit-identical afterwards (same kernels, same arguments -- only the streams differ, so any race shows up here).
    python scripts/soak_side_streams.py [steps] [batch]        (SOAK_DP=1: a one-rank data-parallel step, every collective active)"""
import hashlib, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch, lidog_amd
from lidog_amd import synth
from lidog_amd.trainer import FlatAdam, LiDOGStep
steps, bs = int(sys.argv[1]), int(sys.argv[2])
DP = os.environ.get("SOAK_DP") == "1"      # one-rank RCCL group, SyncBatchNorm + gradient buckets active (as LIDOG_BENCH_SINGLE_RANK_DP=1)
if DP:
    import socket
    import torch.distributed as dist
    import lidog_amd.me as ME
    from lidog_amd.trainer import GradientBuckets, setup_data_parallel
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ME.MinkowskiSyncBatchNorm.single_rank = GradientBuckets.single_rank = True
torch.manual_seed(1234)
model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train()
if DP:
    model = setup_data_parallel(model)
step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
batches = [synth.make_batch(range(bs * i, bs * i + bs), "kitti120k", "cuda") for i in range(2)]
ready = torch.cuda.Event(); ready.record(); torch.cuda.synchronize()
for i in range(steps):
    out = step.training_step(batches[i %% 2], prefetch=batches[(i + 1) %% 2], prefetch_ready=ready)
torch.cuda.synchronize()
h = hashlib.sha1()
for n, t in list(model.named_parameters()) + list(model.named_buffers()):
    h.update(t.detach().cpu().numpy().tobytes())
print("RESULT", h.hexdigest(), float(out["loss"]), getattr(step, "last_path", ""))
''' % R
steps = sys.argv[1] if len(sys.argv) > 1 else "60"
bs = sys.argv[2] if len(sys.argv) > 2 else "4"
res = {}
for name, env in (("side streams on", {}), ("side streams off", {"LIDOG_SIDE_FORWARD": "0", "LIDOG_SIDE_BACKWARD": "0"})):
    p = subprocess.run([sys.executable, "-c", CHILD, steps, bs], env=dict(os.environ, **env), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    if p.returncode != 0 or not line:
        sys.exit(f"{name}: failed\n{p.stderr[-2000:]}")
    res[name] = line[-1]
    print(f"{name}: {line[-1]}", flush=True)
a, b = (r.split()[1] for r in res.values())
print("bit-identical after", steps, "steps:", a == b)
sys.exit(0 if a == b else 1)
