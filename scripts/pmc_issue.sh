#!/bin/bash
# Counter passes that say what the segment kernel's waves wait for: instruction cache, scalar data cache,
# issue activity. usage: scripts/pmc_issue.sh TAG scene:W:H:spp  -> gpurun_out/pmc_issue_TAG.txt
set -e
TAG=$1
SPEC=${2:-Cornell Smoke:1920:1080:64}
OUT=$PWD/gpurun_out/pmc_issue_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH --output-format csv -d "$OUT/p1" -- python3 $ROOT/scripts/perf_few.py "$SPEC" > "$OUT/p1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL --output-format csv -d "$OUT/p2" -- python3 $ROOT/scripts/perf_few.py "$SPEC" > "$OUT/p2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/p3" -- python3 $ROOT/scripts/perf_few.py "$SPEC" > "$OUT/p3.log" 2>&1
cd "$ROOT"
python3 - "$OUT" > gpurun_out/pmc_issue_$TAG.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_segments" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot): print(f"{k:32s} {tot[k]:.4g}")
g = lambda k: tot.get(k, 0.0)
wc = g("SQ_WAVE_CYCLES") or 1
print("icache hit rate", g("SQC_ICACHE_HITS") / (g("SQC_ICACHE_REQ") or 1), "miss", g("SQC_ICACHE_MISSES") / (g("SQC_ICACHE_REQ") or 1))
print("dcache hit rate", g("SQC_DCACHE_HITS") / (g("SQC_DCACHE_REQ") or 1))
print("valu active / wave cycle", g("SQ_ACTIVE_INST_VALU") / wc, " scalar", g("SQ_ACTIVE_INST_SCA") / wc, " wait_inst_any", g("SQ_WAIT_INST_ANY") / wc, " wait_any", g("SQ_WAIT_ANY") / wc)
print("valu insts / salu insts / smem insts", g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_SMEM"))
print("lane utilisation", g("SQ_THREAD_CYCLES_VALU") / (64 * (g("SQ_ACTIVE_INST_VALU") or 1)) if False else "")
PY
cat gpurun_out/pmc_issue_$TAG.txt; tail -1 "$OUT/p1.log"
