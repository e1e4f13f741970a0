run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --precision strict --steps 5 --warmup 2 --no-cpu-baseline --no-batch-sweep --no-resident-leg $EXTRA 2>&1 >/dev/null | grep "images/s" | tail -1; }
run VIP_NOOP=1
run VIP_G8P_MINK=512 VIP_PWK_XLK=512
run VIP_G8P_MINK=512 VIP_PWK_XLK=512 VIP_PWK_WN2K=768
run VIP_G8P_MINK=384 VIP_PWK_XLK=512 VIP_PWK_WN2K=768
run VIP_NOOP=2
EXTRA="--batch 384" run VIP_NOOP=3
EXTRA="--batch 512" run VIP_NOOP=4
