#!/bin/bash
# Profiles the default bench command on the GPU box: kernel trace + stats, then the HBM traffic
# counters in separate counter-only passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), then
# an SQ pass for lane utilisation / wait shares. Output: gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r01}
shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --no-cpu-baseline $*"
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_fetch.json"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_write.json"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_sq.json"
find "$OUT" -name '*.csv' | sort
