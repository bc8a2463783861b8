#!/bin/bash
# Times a few scenes with every library variant ray_tracing_amd/libmrt_<tag>.so given: perf_variants.sh tag...
# ("" = the shipped library). Stops at the first failing variant.
SPECS=("Cornell Smoke:1920:1080:256" "Light Demonstration:1920:1080:64" "Voxel City:3840:2160:8" "Random Scene:800:450:32" "Easy Scene:1280:720:16" "Fast Oct Tree Volume Ice:1920:1080:16" "Fast Oct Tree Sphere:1920:1080:16")
for v in "$@"; do
  lib=$PWD/ray_tracing_amd/libmrt${v:+_$v}.so
  echo "== ${v:-shipped}"
  MRT_LIBRARY=$lib timeout -k 10 200 python scripts/perf_few.py "${SPECS[@]}" 2>/dev/null || exit 1
done
