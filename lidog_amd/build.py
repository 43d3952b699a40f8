"""Builds lidog_amd/_C/liblidog_amd.so (HIP, gfx950 only) in-tree with hipcc.

`python -m lidog_amd.build` or `lidog_amd.build.build()`.  The .so is git-ignored but travels to
the GPU box with the gpurun snapshot; nothing is JIT-compiled at run time."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_C")
SO = os.path.join(OUT_DIR, "liblidog_amd.so")
SOURCES = ["coords.hip", "sconv.hip", "sconv_mfma.hip", "sconv_os.hip", "bn.hip", "bev.hip", "conv2d.hip", "conv2d_sparse.hip", "data.hip", "losses.hip", "optim.hip", "comm.hip", "trunk.hip", "hostprep.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "lidog_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return SO
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src}\n{out.decode()}\n")
        elif verbose and out:
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", SO] + objs + ["-ldl"])
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
