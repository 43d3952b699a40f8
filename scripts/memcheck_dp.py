import os, sys, subprocess, json
env = dict(os.environ, LIDOG_BENCH_SINGLE_RANK_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
code = r'''
import os, sys, torch
sys.argv = ["bench.py", "--steps", "60", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-timing"]
sys.path.insert(0, os.getcwd())
import runpy, atexit
def rep():
    print("MEM max_allocated GB", torch.cuda.max_memory_allocated() / 2**30, "reserved GB", torch.cuda.memory_reserved() / 2**30)
atexit.register(rep)
runpy.run_path("bench.py", run_name="__main__")
'''
r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
print(r.stdout[-700:]); print(r.stderr[-500:])
