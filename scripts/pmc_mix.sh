#!/bin/bash
# Instruction mix of the segment kernel (one counter-only pass).
set -e
OUT=$PWD/gpurun_out/prof_mix
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$OUT/mix" -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 256 > "$OUT/bench.json"
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py "$OUT"/mix/*/*_counter_collection.csv
tail -1 "$OUT/bench.json"
