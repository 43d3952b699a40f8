#!/bin/bash
# first-use flake hunt: the first executor-vs-operator test in a FRESH process, N times per environment
# scripts/flake_first_use.sh <out> <N> "ENV=.." ...
out=$1; n=$2; shift 2
for envs in "$@"; do
  [ "$envs" = "-" ] && envs=""
  pass=0; fail=0
  for i in $(seq 1 $n); do
    if env $envs python -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "bit_identical and True-fused and not output_stationary" > gpurun_out/_flake_run.txt 2>&1; then pass=$((pass+1)); else fail=$((fail+1)); grep -E "^E   .*losses differ|^E  .*differ" gpurun_out/_flake_run.txt | cut -c1-400 >> $out.fails; fi
  done
  echo "$envs: pass $pass fail $fail" >> $out
done
