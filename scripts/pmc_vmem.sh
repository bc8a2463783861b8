#!/bin/bash
# Counter passes for the vector-memory and LDS paths of the segment kernel: how busy the texture addresser and the
# L1 are, how many L1 accesses and L2 requests a launch makes, LDS activity and bank conflicts.
# usage: scripts/pmc_vmem.sh TAG scene:W:H:spp  -> gpurun_out/pmc_vmem_TAG.txt
set -e
TAG=$1
SPEC=${2:-Random Scene:800:450:64}
OUT=$PWD/gpurun_out/pmc_vmem_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/p1" -- python3 $ROOT/scripts/perf_few.py "$SPEC" > "$OUT/p1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2" -- python3 $ROOT/scripts/perf_few.py "$SPEC" > "$OUT/p2.log" 2>&1
cd "$ROOT"
python3 - "$OUT" > gpurun_out/pmc_vmem_$TAG.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.Counter()
n = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_segments" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:40s} {tot[k]:.5g}   ({n[k]} launches)")
g = lambda k: tot.get(k, 0.0)
wc = g("SQ_WAVE_CYCLES") or 1
print("per wave cycle: vmem active", g("SQ_ACTIVE_INST_VMEM") / wc, " lds active", g("SQ_ACTIVE_INST_LDS") / wc)
print("lds: index-active cycles per busy cycle", g("SQ_LDS_IDX_ACTIVE") / (g("SQ_BUSY_CYCLES") or 1), " bank conflict share", g("SQ_LDS_BANK_CONFLICT") / (g("SQ_LDS_IDX_ACTIVE") or 1))
print("L1 accesses per vector load instruction", g("TCP_TOTAL_CACHE_ACCESSES_sum") / (g("SQ_INSTS_VMEM_RD") or 1), " L2 read requests per L1 access", g("TCP_TCC_READ_REQ_sum") / (g("TCP_TOTAL_CACHE_ACCESSES_sum") or 1))
ga = g("GRBM_GUI_ACTIVE") or 1
print("TA busy (average over the addressers) / GPU active", g("TA_BUSY_avr") / ga, " max", g("TA_BUSY_max") / ga)
PY
cat gpurun_out/pmc_vmem_$TAG.txt; tail -1 "$OUT/p1.log"
