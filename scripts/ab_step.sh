#!/bin/bash
# A/B of environment variants in the training step (one stream + two streams kernel time, and bench wall time):
#   bash scripts/ab_step.sh "VAR=a" "VAR=b" ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --no-cpu-baseline --steps 20 --min-seconds 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(r['value'],2), 'scans/s', r['blocks_ms_per_step'])"
done
