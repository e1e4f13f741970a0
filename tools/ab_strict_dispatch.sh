set -e
run() { echo "== $1"; env $1 timeout -k 10 200 python bench.py --precision strict --steps 6 --warmup 2 --no-cpu-baseline --no-batch-sweep --no-resident-leg 2>&1 >/dev/null | grep "images/s"; }
run "VIP_NOOP=1"
run "VIP_G8P_MINK=768"
run "VIP_G8P_MINK=512"
run "VIP_PWK_WN2K=768"
run "VIP_PWK_WN2K=512"
run "VIP_G8P_MINK=768 VIP_PWK_WN2K=768"
run "VIP_PWK_XLK=512"
run "VIP_PWK_XLK=1024"
run "VIP_NOOP=2"
