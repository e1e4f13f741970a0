# A/B timing of dwconv_mfma variants built by tools/build_variant.sh (VIP_DW_SKIP bits: 1 no DMA, 2 no transposition, 4 no MFMA, 8 no stores)
cd $GRAFT_REPO_ROOT
echo "full" >> gpurun_out/dwexp.log
timeout -k 10 120 python tools/bench_dw.py 2>/dev/null > gpurun_out/dwexp_tmp.log; sed -n 1,4p gpurun_out/dwexp_tmp.log >> gpurun_out/dwexp.log
for f in vip-cup-2022_amd/variants/libvipcup_dwskip*.so; do
  echo "$f" >> gpurun_out/dwexp.log
  VIP_LIB_PATH=$PWD/$f timeout -k 10 120 python tools/bench_dw.py 2>/dev/null > gpurun_out/dwexp_tmp.log; sed -n 1,4p gpurun_out/dwexp_tmp.log >> gpurun_out/dwexp.log
done
