#!/bin/bash
# the two PMC passes of tools/run_profiles.sh for the STRICT step only -> gpurun_out/prof/hbm_traffic_pmc_strict.json
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp VIP_STREAMS=1 VIP_BIAS_CALIBRATION=0
OUT=gpurun_out/prof
mkdir -p $OUT
SPMC_ARGS="bench.py --precision strict --steps 2 --warmup 1 --no-cpu-baseline --no-resident-leg --no-batch-sweep --distinct-batches 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_s -- python3 $SPMC_ARGS > $OUT/pmc_fetch_s.json 2> $OUT/pmc_fetch_s.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_s -- python3 $SPMC_ARGS > $OUT/pmc_write_s.json 2> $OUT/pmc_write_s.err
echo write done
python3 tools/pmc_traffic.py $OUT/pmc_fetch_s $OUT/pmc_write_s $OUT/hbm_traffic_pmc_strict.json > $OUT/pmc_traffic_strict.log 2>&1
rm -rf $OUT/pmc_fetch_s $OUT/pmc_write_s
tail -30 $OUT/pmc_traffic_strict.log
