#!/bin/bash
# Stall attribution for the matrix kernels:  bash scripts/profile_stalls.sh <tag> [BS]
# Six rocprofv3 --pmc passes (two more defined, see below) over scripts/prof_train.py (2 training steps, one stream: counters serialise the
# kernels anyway), each within the per-block slot limits of gfx950 (SQ 8, TCC 4, GRBM 2; MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  Counters in their own passes, never with a trace domain; the program directly after `--`.
# A pass whose counter set the profiler refuses is skipped (it exits at once); a pass that is KILLED stops the script.
# Summary: gpurun_out/profiles/<tag>_stalls_*.txt  (copy into profiles/).
tag=$1; bs=${2:-4}
cd "$(dirname "$0")/.."
out=gpurun_out/profiles; mkdir -p $out
export TMPDIR=/tmp LIDOG_BACKWARD_OVERLAP=0 BS=$bs STEPS=2
declare -A P
P[wait]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
P[active]="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
P[insts]="SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P[level]="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_IFETCH"
P[l2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE"
P[ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr"
P[tcp]="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum"
P[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum"
dirs=""
# (round 6: the "tcp" pass -- TCP_TCC_READ_REQ_LATENCY & co -- never finished on this pool and was killed at its limit, which
# also kept "ta" from running; both stay defined above but are NOT run by default:  PASSES="... tcp ta" to try again)
for name in ${PASSES:-wait active insts level l2 ea}; do
  d=gpurun_out/stalls_${tag}_$name; rm -rf $d
  echo "== pass $name: ${P[$name]}"
  timeout -k 10 420 rocprofv3 --pmc ${P[$name]} --output-format csv -d $d -o p -- python3 scripts/prof_train.py > gpurun_out/stalls_${tag}_$name.log 2>&1
  rc=$?
  echo "   rc $rc"
  if [ $rc -ge 124 ]; then echo "pass $name killed (rc $rc): stopping"; break; fi
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/stalls_${tag}_$name.log; continue; fi
  dirs="$dirs $d"
done
python3 scripts/pmc_stalls.py $dirs > $out/${tag}_stalls_bs${bs}.txt 2> $out/${tag}_stalls_bs${bs}.err
python3 scripts/pmc_stalls.py --detail $dirs > $out/${tag}_stalls_bs${bs}_by_shape.txt 2>/dev/null
for d in $dirs; do rm -rf $d; done
head -60 $out/${tag}_stalls_bs${bs}.txt
