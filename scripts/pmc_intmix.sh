#!/bin/bash
# Integer / conversion / f32 share of the segment kernel's vector instructions (one counter-only pass).
# usage: scripts/pmc_intmix.sh [bench.py arguments]
set -e
OUT=$PWD/gpurun_out/prof_intmix
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/mix" -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 256 $* > "$OUT/bench.json"
python3 $ROOT/scripts/pmc_summary.py "$OUT"/mix/*/*_counter_collection.csv
