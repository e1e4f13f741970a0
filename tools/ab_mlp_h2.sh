for NWV in 4 8; do echo "== VIP_MLP_H2_WAVES=$NWV"; VIP_MLP_H2_WAVES=$NWV timeout -k 10 200 python tools/bench_mlp_h2.py 2>/dev/null; done
