#!/bin/bash
# PMC of one dense shape of the pointwise-GEMM family:  tools/pmc_gemm.sh M N K act tag   -> gpurun_out/pmc_gemm/<tag>.txt
# Counters in separate passes (no tracing flags beside --pmc).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
M=$1; N=$2; K=$3; ACT=$4; TAG=$5
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
python3 tools/bench_gemm_one.py $M $N $K $ACT 20 > $OUT/$TAG.txt 2>&1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  rocprofv3 --pmc $set --output-format csv -d $OUT/p_$i -- python3 tools/bench_gemm_one.py $M $N $K $ACT 3 > /dev/null 2> $OUT/p_$i.err
  python3 tools/pmc_parse.py $OUT/p_$i pw >> $OUT/$TAG.txt 2>&1
  rm -rf $OUT/p_$i
  i=$((i+1))
done
cat $OUT/$TAG.txt | grep -v "^void\|^_ZN" 
