#!/bin/bash
# Achieved shader clock per matrix-kernel family: diagnostic variants of ONE source file each
# (-DLIDOG_CLOCK_STAMP, csrc/clock_stamp.h) through scripts/build_variant.sh, then scripts/clock_stamps.py.
#   bash scripts/clock_stamps.sh <tag>     ->  gpurun_out/profiles/<tag>_clock_stamps.txt
tag=$1
cd "$(dirname "$0")/.."
out=gpurun_out/profiles; mkdir -p $out
f=$out/${tag}_clock_stamps.txt; : > $f
for pair in "gemm sconv_mfma.hip" "wgrad sconv_mfma.hip" "os sconv_os.hip" "conv2d conv2d.hip"; do
  set -- $pair
  so=$(bash scripts/build_variant.sh stamp_$1 $2 -DLIDOG_CLOCK_STAMP | tail -1)
  echo "== $1 ($2)" | tee -a $f
  LIDOG_SO=$PWD/$so timeout -k 10 300 python scripts/clock_stamps.py $1 2>&1 | grep -v amdgpu.ids | tee -a $f
  rc=${PIPESTATUS[0]}
  if [ $rc -ge 124 ]; then echo "killed (rc $rc): stopping" | tee -a $f; break; fi
done
