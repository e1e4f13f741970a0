# A/B of kernel-library variants (tools/build_variant.sh) under the concurrency race test + the attention micro-benchmark
mkdir -p gpurun_out
for tag in ${TAGS:-default}; do
  if [ $tag = default ]; then unset VIP_LIB_PATH; else export VIP_LIB_PATH=$PWD/vip-cup-2022_amd/variants/libvipcup_$tag.so; fi
  echo "== $tag" >> gpurun_out/race_variants.log
  timeout -k 10 150 python tools/race_matrix.py --iters 40 --victims attn14,attn7 --aggressors mfma_only,pwk_plain,pw_stream,pwk_gelu,conv3x3 2>&1 | grep race2 >> gpurun_out/race_variants.log
  timeout -k 10 100 python tools/bench_attn.py 2>&1 | grep "^L" >> gpurun_out/race_variants.log
done
cat gpurun_out/race_variants.log
