set -e
b() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],2))"; }
b "prio=1 tile=1"
b "prio=1 tile=1"
touch vip-cup-2022_amd/csrc/conv_igemm.hip
VIP_EXTRA_CXXFLAGS="-DVIP_MFMA_PRIO=0 -DVIP_MFMA_PRIO_TILE=0" python vip-cup-2022_amd/build.py > /dev/null
b "prio=0 tile=0"
b "prio=0 tile=0"
touch vip-cup-2022_amd/csrc/conv_igemm.hip
VIP_EXTRA_CXXFLAGS="-DVIP_MFMA_PRIO=1 -DVIP_MFMA_PRIO_TILE=0" python vip-cup-2022_amd/build.py > /dev/null
b "prio=1 tile=0"
VIP_STREAMS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial prio=1 tile=0', round(d['ms_per_step'],2))"
