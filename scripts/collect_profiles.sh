#!/bin/bash
# Copies the summaries of scripts/profile_bench.sh runs (gpurun_out/prof_<tag>/) into profiles/ (tracked). The bench
# lines that quote them (profiles/<tag>_bench_line.json) come from scripts/bench_lines.sh, run afterwards.
# usage: scripts/collect_profiles.sh tag...
set -e
for tag in "$@"; do
  src=gpurun_out/prof_$tag
  cp "$src/kernel_stats.csv" "profiles/${tag}_kernel_stats.csv"
  cp "$src/summary.json" "profiles/${tag}_pmc_traffic.json"
  echo "collected $tag"
done
