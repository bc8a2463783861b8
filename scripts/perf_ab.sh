#!/bin/bash
# Times given scene specs with every library variant named: perf_ab.sh "spec;spec" tag... ("" = the shipped library)
IFS=';' read -ra SPECS <<< "$1"; shift
for v in "$@"; do
  lib=$PWD/ray_tracing_amd/libmrt${v:+_$v}.so
  echo "== ${v:-shipped}"
  MRT_LIBRARY=$lib timeout -k 10 200 python scripts/perf_few.py "${SPECS[@]}" 2>/dev/null || exit 1
done
