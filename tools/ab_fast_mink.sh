run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batch-sweep --no-resident-leg --no-strict-leg 2>&1 >/dev/null | grep "images/s" | tail -1; }
run VIP_NOOP=1
run VIP_G8P_MINK=384
run VIP_NOOP=2
run VIP_G8P_MINK=384
run VIP_G8P_MINK=768
