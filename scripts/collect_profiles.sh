#!/bin/bash
# Copies the summaries of scripts/profile_bench.sh runs (gpurun_out/prof_<tag>/) into profiles/ (tracked).
# usage: scripts/collect_profiles.sh tag...
set -e
for tag in "$@"; do
  src=gpurun_out/prof_$tag
  cp "$src/kernel_stats.csv" "profiles/${tag}_kernel_stats.csv"
  cp "$src/summary.json" "profiles/${tag}_pmc_traffic.json"
  tail -1 "$src/bench_trace.json" > "profiles/${tag}_bench_line.json"
  echo "collected $tag"
done
