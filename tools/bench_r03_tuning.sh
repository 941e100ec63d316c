#!/bin/bash
# zero-code tuning sweeps (round 3): sequential solves with the per-phase breakdown
B="python bench.py --steps 4 --warmup 1 --cpu-sample 0 --in-flight 1"
mkdir -p gpurun_out
$B > gpurun_out/t_default.json 2>/dev/null
for v in 8192 16384 32768; do $B --team-threshold $v > gpurun_out/t_team$v.json 2>/dev/null; done
for v in 60 85 92; do $B --compact-percent $v > gpurun_out/t_compact$v.json 2>/dev/null; done
for v in 2048 4096; do $B --wave-threshold $v > gpurun_out/t_wave$v.json 2>/dev/null; done
$B --spec-threshold 16384 > gpurun_out/t_spec16k.json 2>/dev/null
$B --spec-threshold 4096 > gpurun_out/t_spec4k.json 2>/dev/null
