"""Secondary workloads of BASELINE.json (not the headline bench): training-step throughput, one GPU, of
  C1  train_source.py  MinkUNet34, 8 k-point scans, 0.1 m voxels, bs 4                         (configs[0])
  C4  train_aug_based.py  MinkUNet34 on Mix3D nuScenes-like unions of two scans, bs 4          (configs[3])
  C5  train_lidog.py  MinkUNet34 + BEV head, 128 x 4096-beam scans at 0.02 m, bs 1             (configs[4])
one JSON line each, with the same `roofline` fields as bench.py's line: the gathered GEMM's launches timed by HIP events
inside the executor (`frac` of the f32 matrix peak) and the whole step's dense-equivalent FLOPs / algorithmic bytes
(scripts/count_work.py on scan seed 0 of each config: SURVEY.md 8(d) formulas; a training step = 3 x forward) against the
fp32 and HBM roofs.      python scripts/bench_configs.py [steps]"""
import ctypes, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import lidog_amd
from lidog_amd import synth, _lib
from lidog_amd.trainer import FlatAdam, LiDOGStep, SourceStep

FP32_PEAK_TFLOPS, HBM_PEAK_GBS = 157.3, 8000.0
# per scan, forward (python scripts/count_work.py <config> 0 [mix3d]): sparse GFLOP, compulsory GB; + BEV head 82.1 GFLOP / 0.78 GB
WORK = {"source8k": (23.2, 0.55, False), "nusc35k": (212.2, 3.34, False), "highres524k": (1307.7 + 82.1, 19.2, True)}

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = _lib.load()
for name, cfg, mix, bs in (("C1 source8k", "source8k", False, 4), ("C4 mix3d nusc35k", "nusc35k", True, 4),
                           ("C5 highres524k", "highres524k", False, 1)):
    torch.manual_seed(0)
    gflop, gb, bev = WORK[cfg]
    if bev:
        model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train()
        step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
    else:
        model = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
        step = SourceStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
    batches = [synth.make_batch(range(bs * i, bs * i + bs), cfg, "cuda", mix3d=mix) for i in range(2)]
    nvox = sum(b["coords_int"].shape[0] for b in batches) / (2 * bs)
    ready = torch.cuda.Event(); ready.record(); torch.cuda.synchronize()
    for i in range(3):
        step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)
    torch.cuda.synchronize()
    L.lidog_trunk_gemm_timing(0)
    buf = (ctypes.c_double * 4)()
    L.lidog_trunk_gemm_timing_read(buf)          # clears what the warm-up may have left
    t0 = time.perf_counter()
    timed = 0
    for i in range(steps):
        on = i % 10 == 0                          # as bench.py: events around the GEMM launches on every 10th step
        L.lidog_trunk_gemm_timing(1 if on else 0)
        timed += int(on)
        out = step.training_step(batches[(i + 1) % 2], prefetch=batches[i % 2], prefetch_ready=ready)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.lidog_trunk_gemm_timing(0)
    L.lidog_trunk_gemm_timing_read(buf)
    launches, ms, flops, bytes_ = buf[0], buf[1], buf[2], buf[3]
    step_s = dt / steps
    line = {"metric": "LiDAR scans/sec, training step", "config": {"workload": f"{name}: {nvox:.0f} voxels/scan, bs {bs}"
                                                                   f"{', MinkUNet34 + BEV head' if bev else ', MinkUNet34'}"},
            "value": bs * steps / dt, "unit": "scans/s", "ms_per_step": 1e3 * step_s, "steps": steps, "loss": float(out["loss"]),
            "trunk_path": getattr(step, "last_path", "")}
    if launches:
        tfl = flops / (ms * 1e-3) / 1e12
        line["roofline"] = {"bound": "mfma", "kernel": "k_sconv_gemm_mfma", "achieved": tfl, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tfl / FP32_PEAK_TFLOPS, "launches": int(launches), "avg_launch_us": 1e3 * ms / launches,
                            "hbm_gbs": bytes_ / (ms * 1e-3) / 1e9, "instrumented_steps": timed,
                            "step_dense_equivalent_gflop": 3 * gflop * bs, "step_algorithmic_gb": 3 * gb * bs,
                            "step_fp32_frac": 3 * gflop * bs / step_s / 1e3 / FP32_PEAK_TFLOPS,
                            "step_hbm_frac": 3 * gb * bs / step_s / HBM_PEAK_GBS}
    print(json.dumps(line), flush=True)
    del model, step, batches
    torch.cuda.empty_cache()
