#!/bin/bash
# one-handle pipelining experiments (round 3): python bench.py variants, one JSON line each into gpurun_out/v_*.json
B="python bench.py --steps 6 --warmup 2 --cpu-sample 0"
mkdir -p gpurun_out
$B > gpurun_out/v_default.json 2>/dev/null
CILQR_FIN_PRIORITY=0 $B > gpurun_out/v_prio0.json 2>/dev/null
$B --finish-threshold 4096 > gpurun_out/v_fin4096.json 2>/dev/null
$B --finish-threshold 2048 > gpurun_out/v_fin2048.json 2>/dev/null
$B --tail-threshold 512 > gpurun_out/v_tail512.json 2>/dev/null
$B --tail-threshold 1024 > gpurun_out/v_tail1024.json 2>/dev/null
$B --batch 32768 --pipeline 2 > gpurun_out/v_half2.json 2>/dev/null
$B --batch 16384 --pipeline 4 > gpurun_out/v_quarter4.json 2>/dev/null
$B --batch 32768 --pipeline 3 > gpurun_out/v_half3.json 2>/dev/null
