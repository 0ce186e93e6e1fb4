#!/bin/bash
# PMC passes over one lab variant on one shape: ./run_pmc.sh <variant-substr> <M> <N> <K> <tag>
set -e
cd /tmp && export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/tools/gemm_lab/gemm_lab
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$5
mkdir -p $OUT
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- $LAB 3 "$1" $2 $3 $4 > $OUT/p1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- $LAB 3 "$1" $2 $3 $4 > $OUT/p2.log 2>&1
