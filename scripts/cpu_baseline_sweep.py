"""thread sweep of bench.py's cpu_baseline leg (run on the GPU box's host: 128 cores)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
for t in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    r = bench.cpu_baseline("kitti120k", steps_budget_s=1.0, threads=t)
    print(t, "threads:", round(1 / r["value"], 2), "s/scan", flush=True)
