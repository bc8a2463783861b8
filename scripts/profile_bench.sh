#!/bin/bash
# Profiles one bench command on the GPU box: kernel trace + stats, then counter-only passes (each
# in a run of its own): HBM traffic (FETCH_SIZE and WRITE_SIZE do not fit one pass), an SQ pass for
# lane utilisation / VALU activity / wait shares, and the f64 instruction mix.
# usage: [MRT_PROFILE_WAVES=2] scripts/profile_bench.sh TAG [bench.py arguments without spaces inside an argument]
# (MRT_PROFILE_WAVES: resident waves per SIMD of the profiled kernel, for the issue-occupancy figure; default 3)
# Output: gpurun_out/prof_TAG/ and its summary gpurun_out/prof_TAG/summary.json (scripts/pmc_traffic.py).
set -e
TAG=${1:-r02}
shift || true
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
BENCH="python3 $ROOT/bench.py --no-cpu-baseline $*"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_fetch.json"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_write.json"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_sq.json"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$OUT/pmc_mix" -- $BENCH > "$OUT/bench_mix.json"
cd "$ROOT"
python3 scripts/pmc_traffic.py "$OUT" "$OUT/summary.json" ${MRT_PROFILE_WAVES:-3} > /dev/null
cp "$OUT"/trace/*/*_kernel_stats.csv "$OUT/kernel_stats.csv"
echo "profile $TAG done"
