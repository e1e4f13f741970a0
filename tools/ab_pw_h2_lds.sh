for KB in 72 112 156; do
  echo "== VIP_PW_H2_LDS_KB=$KB"
  VIP_PW_H2_LDS_KB=$KB VIP_PRECISION=strict timeout -k 10 200 python tools/profile_shapes.py ensemble8 256 60 2>/dev/null | grep "total instr\|pw_gemm_kernel "
  VIP_PW_H2_LDS_KB=$KB timeout -k 10 200 python bench.py --precision strict --steps 6 --warmup 2 --no-cpu-baseline --no-batch-sweep --no-resident-leg 2>&1 >/dev/null | grep "images/s"
done
