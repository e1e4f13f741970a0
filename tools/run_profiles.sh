#!/bin/bash
# rocprofv3 passes behind bench.py's roofline numbers (run on the GPU box from the repo root; outputs under gpurun_out/prof/):
#   1. kernel trace + stats of the bench command as timed (3 member streams)            -> kernel_stats_streams.csv
#   2. the same with VIP_STREAMS=1 (one stream: per-kernel durations are not inflated)    -> kernel_stats_serial.csv
#   3. + 4. PMC FETCH_SIZE / WRITE_SIZE in separate passes (VIP_STREAMS=1)                -> tools/pmc_traffic.py -> hbm_traffic_pmc.json
# VIP_BIAS_CALIBRATION=0 keeps the one-off 16-image calibration launches out of the averages.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-resident-leg --no-strict-leg --no-batch-sweep --distinct-batches 4"
export VIP_BIAS_CALIBRATION=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/streams -- python3 $ARGS > $OUT/bench_streams.json 2> $OUT/bench_streams.err
cp $(find $OUT/streams -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_streams.csv
export VIP_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $ARGS > $OUT/bench_serial.json 2> $OUT/bench_serial.err
cp $(find $OUT/serial -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_serial.csv
# 2b. the STRICT precision step (packed fp16-pair storage): serial and as timed                -> kernel_stats_strict{,_streams}.csv
SARGS="bench.py --precision strict --steps 5 --warmup 2 --no-cpu-baseline --no-resident-leg --no-batch-sweep --distinct-batches 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/strict_serial -- python3 $SARGS > $OUT/bench_strict_serial.json 2> $OUT/bench_strict_serial.err
cp $(find $OUT/strict_serial -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_strict.csv
unset VIP_STREAMS
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/strict_streams -- python3 $SARGS > $OUT/bench_strict_streams.json 2> $OUT/bench_strict_streams.err
cp $(find $OUT/strict_streams -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_strict_streams.csv
rm -rf $OUT/strict_serial $OUT/strict_streams
export VIP_STREAMS=1
PMC_ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-resident-leg --no-strict-leg --no-batch-sweep --distinct-batches 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $PMC_ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $PMC_ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/hbm_traffic_pmc.json > $OUT/pmc_traffic.log 2>&1
# 5. + 6. the same two counters over the STRICT step                                            -> hbm_traffic_pmc_strict.json
SPMC_ARGS="bench.py --precision strict --steps 2 --warmup 1 --no-cpu-baseline --no-resident-leg --no-batch-sweep --distinct-batches 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_s -- python3 $SPMC_ARGS > $OUT/pmc_fetch_s.json 2> $OUT/pmc_fetch_s.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_s -- python3 $SPMC_ARGS > $OUT/pmc_write_s.json 2> $OUT/pmc_write_s.err
python3 tools/pmc_traffic.py $OUT/pmc_fetch_s $OUT/pmc_write_s $OUT/hbm_traffic_pmc_strict.json > $OUT/pmc_traffic_strict.log 2>&1
rm -rf $OUT/pmc_fetch_s $OUT/pmc_write_s
# the raw traces are large: keep the summaries only
rm -rf $OUT/streams $OUT/serial $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
