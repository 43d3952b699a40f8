#!/bin/bash
# SQ counters of one gathered-GEMM layer: bash scripts/pmc_gemm.sh <S> <CIN> <COUT> <tag>
export TMPDIR=/tmp S=$1 CIN=$2 COUT=$3
tag=$4
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  d=gpurun_out/pmcg_$tag; rm -rf $d
  rocprofv3 --pmc $set --output-format csv -d $d -o p -- python3 scripts/prof_gemm_case.py > gpurun_out/pmcg_$tag.log 2>&1
  python3 scripts/pmc_sq.py $d --filter=k_sconv_gemm
  rm -rf $d
done
