#!/bin/bash
# One round's profile set on the GPU box:  bash scripts/profile_round.sh <tag> [what...]
#   what: trace (kernel trace, two-stream + one-stream), pmc (FETCH_SIZE / WRITE_SIZE / SQ busy passes)   default: both
# Writes summaries straight into profiles/<tag>_* copies under gpurun_out/profiles/ (gpurun merges gpurun_out back);
# copy the ones to keep into profiles/.  CONFIG / BS in the environment select another workload (config 5:
# CONFIG=highres524k BS=1).  The profiled command is scripts/prof_train.py (training steps only), the
# program directly after `--` as gpurun requires; counters in their own passes (never with a trace domain).
set -e
tag=$1; shift
what="${*:-trace pmc}"
cd "$(dirname "$0")/.."
out=gpurun_out/profiles; mkdir -p $out
export TMPDIR=/tmp
STEPS_TRACE=8
if [[ $what == *trace* ]]; then
  for mode in two one; do
    d=gpurun_out/prof_${tag}_$mode; rm -rf $d
    if [ $mode = one ]; then export LIDOG_BACKWARD_OVERLAP=0; else unset LIDOG_BACKWARD_OVERLAP; fi
    STEPS=$STEPS_TRACE rocprofv3 --kernel-trace --stats -d $d -o t -- python3 scripts/prof_train.py > gpurun_out/prof_${tag}_$mode.log 2>&1
    sfx=""; [ $mode = one ] && sfx="_one_stream"
    python3 scripts/kernel_breakdown.py $d/t_results.db $STEPS_TRACE --csv $out/${tag}_kernel_stats_train_bs${BS:-4}$sfx.csv --top > $out/${tag}_kernel_breakdown_train_bs${BS:-4}$sfx.txt
    python3 scripts/step_timeline.py $d/t_results.db > $out/${tag}_step_timeline$sfx.txt 2>&1 || true
    rm -rf $d
  done
  unset LIDOG_BACKWARD_OVERLAP
fi
if [[ $what == *pmc* ]]; then
  export LIDOG_BACKWARD_OVERLAP=0   # counters serialise kernels anyway; one stream keeps launch counts simple
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmc_${tag}_$c; rm -rf $d
    STEPS=2 rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 scripts/prof_train.py > gpurun_out/pmc_${tag}_$c.log 2>&1
  done
  d=gpurun_out/pmc_${tag}_SQ; rm -rf $d
  STEPS=2 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $d -o p -- python3 scripts/prof_train.py > gpurun_out/pmc_${tag}_SQ.log 2>&1 || \
  STEPS=2 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d -o p -- python3 scripts/prof_train.py > gpurun_out/pmc_${tag}_SQ.log 2>&1
  for k in k_sconv_gemm_mfma k_sconv_os_mfma k_sconv_wgrad_mfma k_conv_s2 k_sconv_reduce_rows4; do
    python3 scripts/pmc_traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE $k > $out/${tag}_pmc_traffic_${k#k_}.json 2>/dev/null || rm -f $out/${tag}_pmc_traffic_${k#k_}.json
  done
  python3 scripts/pmc_sq.py gpurun_out/pmc_${tag}_SQ > $out/${tag}_pmc_sq_all_kernels.txt
  python3 scripts/pmc_mfma_busy.py gpurun_out/pmc_${tag}_SQ > $out/${tag}_pmc_mfma_busy.txt
  rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/pmc_${tag}_SQ
fi
ls -la $out
