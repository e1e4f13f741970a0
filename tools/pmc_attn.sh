#!/bin/bash
# PMC comparison of the two ws-14 window-attention kernels (one-item vs persistent LDS-DMA pipeline) on GCViT level 2, B = 256.
# Counters in separate passes of at most a handful each (no tracing flags beside --pmc).  Output: gpurun_out/pmc_attn/*.txt
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp ITERS=3
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
for v in 0 1; do
  export VIP_ATTN_PIPE=$v
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU" \
             "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"; do
    rocprofv3 --pmc $set --output-format csv -d $OUT/p${v}_$i -- python3 tools/bench_attn_l2.py > /dev/null 2> $OUT/p${v}_$i.err
    python3 tools/pmc_parse.py $OUT/p${v}_$i window_attn >> $OUT/pipe$v.txt 2>&1
    rm -rf $OUT/p${v}_$i
    i=$((i+1))
  done
done
tail -n 40 $OUT/pipe0.txt $OUT/pipe1.txt
