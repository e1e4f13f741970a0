#!/bin/bash
# Builds the round-2 window-attention kernel (tools/repro/window_attn_round2.hip) in each bisecting variant into a library of its own
# (the rest of the library as built) and runs the victim x aggressor race matrix on it.  On the GPU box, from the repo root:
#   bash tools/repro/run_race_repro.sh > gpurun_out/race_repro.log 2>&1
cd "$(dirname "$0")/../.." || exit 1
P=vip-cup-2022_amd
mkdir -p $P/variants
for V in ${RACE_VARIANTS:-0 1 3 4 5 6 7 8}; do
  OBJ=/tmp/wa_r2_$V.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$P/csrc -Wno-unused-result -ffp-contract=fast \
      -DVIP_BUILD_EXPERIMENTS=0 -DRACE_VARIANT=$V -x hip -c tools/repro/window_attn_round2.hip -o $OBJ 2>/dev/null || { echo "variant $V: build failed"; continue; }
  OBJS=$(ls $P/build/*.o | grep -v "/window_attn.hip.o" | grep -v "\.exp\.o" | grep -v "gcvit_block14\|dwconv_mfma\|mbconv_fused")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/variants/libvipcup_race$V.so $OBJS $OBJ -lpthread || continue
  echo "== RACE_VARIANT $V"
  VIP_LIB_PATH=$P/variants/libvipcup_race$V.so timeout -k 10 120 python tools/race_matrix.py --iters 40 --victims attn14,attn7,attn14g --aggressors mfma_only,pwk_plain
done
echo "== the shipped kernel (bias table as the MFMA C operand)"
timeout -k 10 120 python tools/race_matrix.py --iters 40 --victims attn14,attn7,attn14g --aggressors mfma_only,pwk_plain
