# phases of the LDS-staged depthwise kernel on ConvNeXt's first 7x7 layer (VIP_DW_LDS_DBG: 1 no math, 2 no stores, 4 no loads)
for D in 0 1 2 4 3 5 6 7; do
  echo "== DBG=$D"; VIP_DW_LDS_DBG=$D ONLY=cnx timeout -k 10 100 python tools/bench_dw_h2.py 2>/dev/null | grep "cnx"
done
