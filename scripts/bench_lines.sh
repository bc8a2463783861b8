#!/bin/bash
# The four configurations' bench lines without the profiler (they quote the counters in profiles/): run after
# scripts/collect_profiles.sh, on the GPU box. Writes gpurun_out/bl_<scene>.json; copy them to profiles/r03_<scene>_bench_line.json.
set -e
mkdir -p gpurun_out
timeout -k 10 120 python bench.py --no-cpu-baseline > gpurun_out/bl_cornell.json
timeout -k 10 120 python bench.py --no-cpu-baseline --scene voxel_city --width 3840 --height 2160 --spp 16 --steps 5 --warmup 1 > gpurun_out/bl_voxel_city.json
timeout -k 10 120 python bench.py --no-cpu-baseline --scene light_demonstration --width 1920 --height 1080 --spp 256 --steps 5 --warmup 1 > gpurun_out/bl_light_demonstration.json
timeout -k 10 120 python bench.py --no-cpu-baseline --scene random_scene --width 800 --height 450 --spp 64 --steps 20 --warmup 2 > gpurun_out/bl_random_scene.json
