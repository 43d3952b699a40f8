#!/bin/bash
# A/B runs of bench.py under different environments: scripts/ab_bench.sh <tag> <steps> "ENV1=.. ENV2=.." "ENV=.." ...
# (AB_ARGS="--batch 2" adds bench arguments; "-" = no extra environment).  One JSON line per run in gpurun_out/ab_<tag>.log, summarised at the end.
tag=$1; steps=$2; shift 2
out=gpurun_out/ab_$tag.log; : > $out
for rep in 1 2; do
  for envs in "$@"; do
    [ "$envs" = "-" ] && envs=""
    line=$(env $envs python bench.py --no-cpu-baseline --steps $steps $AB_ARGS 2>/dev/null | grep '^{' | tail -1)
    echo "{\"env\": \"$envs\", \"rep\": $rep, \"res\": ${line:-null}}" >> $out
  done
done
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["res"]
    if r is None:
        print("%-60s FAILED" % d["env"]); continue
    rf = r.get("roofline") or {}
    print("%-60s rep%d %7.2f scans/s %7.2f ms  gemm frac %.3f" % (d["env"] or "-", d["rep"], r["value"], r["ms_per_step"], rf.get("frac", 0)))
PY
