#!/bin/bash
# kernel trace of the one-rank data-parallel step (real RCCL, every collective active):  bash scripts/profile_dp.sh <tag>
set -e
tag=$1
cd "$(dirname "$0")/.."
out=gpurun_out/profiles; mkdir -p $out
export TMPDIR=/tmp
d=gpurun_out/prof_${tag}_dp; rm -rf $d
DP=1 STEPS=8 rocprofv3 --kernel-trace --stats -d $d -o t -- python3 scripts/prof_train.py > gpurun_out/prof_${tag}_dp.log 2>&1
python3 scripts/kernel_breakdown.py $d/t_results.db 8 --csv $out/${tag}_kernel_stats_train_bs4_dp1.csv --top > $out/${tag}_kernel_breakdown_train_bs4_dp1.txt
python3 scripts/step_timeline.py $d/t_results.db > $out/${tag}_step_timeline_dp1.txt 2>&1 || true
rm -rf $d
