for v in 3 2 3 2; do
  VIP_PW=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VIP_PW=$v', round(d['value'],1), round(d['ms_per_step'],2))"
done
for v in 3 2; do VIP_PW=$v python tools/profile_shapes.py ensemble 256 2>/dev/null | grep "total conv" | sed "s/^/VIP_PW=$v /"; done
