#!/bin/bash
# scripts/ab_kernels.sh <script.py> <variant> [<variant> ...]: run a kernel micro-benchmark against the default build ("-") and variant builds
script=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then so=""; else so="lidog_amd/_C/liblidog_amd_$v.so"; fi
  echo "=== variant: ${v}"
  LIDOG_SO=$so python $script 2>&1 | grep -v amdgpu.ids
done
