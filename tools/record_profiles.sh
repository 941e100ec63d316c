#!/bin/bash
# Records the measurement set of one round on the GPU box into gpurun_out/<tag>/ :
#   bench JSON of the default command; the same command under rocprofv3 --kernel-trace --stats, once as it is (two
#   handles with two solves in flight each: what `value` is measured on) and once with --in-flight 1 (one batch at a time: per-kernel
#   durations that other streams do not inflate, and the per-iteration timeline); PMC passes for the HBM traffic of the
#   backward kernels (separate runs, counters only) on every bench workload; SQ counter passes over one solve for the
#   other kernels; the other BASELINE configs, the DP scene source (the host-array and end-to-end figures are legs of the
#   default bench line since round 6: pcie_inclusive, end_to_end).
#   usage (through gpurun): bash tools/record_profiles.sh r02
# then copy gpurun_out/<tag>/*.{json,txt} into profiles/ (see DESIGN.md "Measurement").
set -u
tag=${1:-r06}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
python bench.py > "$out/${tag}_bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats -d "$out/kt3" -- python bench.py --no-extras --cpu-sample 0 --no-latency > "$out/${tag}_bench_under_rocprofv3.json" 2> "$out/kt3.err"
python tools/prof_summary.py "$out/kt3" --bench-json "$out/${tag}_bench_under_rocprofv3.json" > "$out/${tag}_kernel_stats_pipelined.txt" 2>&1
rm -rf "$out/kt3"   # raw captures go as soon as they are summarised: a call that is cut off must not leave them for the merge (64 MiB cap)
rocprofv3 --kernel-trace --stats -d "$out/kt1" -- python bench.py --no-extras --steps 6 --warmup 2 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency > "$out/${tag}_bench_pipeline1_under_rocprofv3.json" 2> "$out/kt1.err"
{ python tools/prof_summary.py "$out/kt1" --iters 1,2,5,10,20,40,80; python tools/phase_summary.py "$out/kt1"; } > "$out/${tag}_kernel_stats.txt" 2>&1
rm -rf "$out/kt1"
targs=""
for wl in "mix11:50:" "dyn20:100:--scene dyn20" "dyn20x:100:--scene dyn20x"; do
  IFS=: read -r name n extra <<< "$wl"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$out/pmc_${name}_$c" -- python bench.py --no-extras --steps 1 --warmup 0 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency $extra > "$out/pmc_${name}_$c.json" 2> "$out/pmc_${name}_$c.err"
    python tools/pmc_kernel.py "$out/pmc_${name}_$c" k_backward > "$out/${tag}_pmc_backward_${name}_$c.txt" 2>&1
    rm -rf "$out/pmc_${name}_$c"
  done
  targs="$targs ${name}_n${n} $out/${tag}_pmc_backward_${name}_FETCH_SIZE.txt $out/${tag}_pmc_backward_${name}_WRITE_SIZE.txt $out/pmc_${name}_FETCH_SIZE.json"
done
python tools/make_traffic_json.py "$out/backward_traffic.json" $targs > "$out/traffic.log" 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d "$out/pmc_all_$i" -- python bench.py --no-extras --steps 1 --warmup 0 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency > "$out/pmc_all_$i.json" 2> "$out/pmc_all_$i.err"
  python tools/pmc_summary.py "$out/pmc_all_$i" > "$out/${tag}_pmc_all_kernels_$i.txt" 2>&1
  rm -rf "$out/pmc_all_$i"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/r2f" -- python bench.py --no-extras --steps 1 --warmup 0 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency --no-traffic > /dev/null 2> "$out/r2f.err"
python tools/round2_fetch.py "$out/r2f" > "$out/${tag}_linesearch_round2_fetch.json" 2>> "$out/r2f.err"
rm -rf "$out/r2f"
python tools/kernel_rooflines.py "$out" "$tag" > "$out/${tag}_kernel_rooflines.json" 2> "$out/kr.err"
CILQR_BENCH_FORCE_DIST=1 python bench.py --no-extras --cpu-sample 0 --no-latency > "$out/${tag}_bench_force_dist.json" 2> "$out/fd.err"
CILQR_BENCH_FORCE_DIST=1 python bench.py --no-extras --gather c_abi --cpu-sample 0 --no-latency > "$out/${tag}_bench_force_dist_c_abi_gather.json" 2> "$out/fdc.err"
CILQR_BENCH_MULTI_DEVICES=0,0 python bench.py --no-extras --gpus 2 --multi --cpu-sample 0 --no-latency > "$out/${tag}_bench_multi_two_shards_one_gpu.json" 2> "$out/mu.err"
# host CPU of the timed region (round 5): the default (submitted solves nap in their host waits) against spinning waits
CILQR_HOST_WAIT=spin python bench.py --no-extras --cpu-sample 0 --no-latency > "$out/${tag}_bench_host_wait_spin.json" 2> "$out/spin.err"
python tools/bwd_forms_sweep.py > "$out/${tag}_backward_forms_sweep.json" 2> "$out/forms.err"
bash tools/tsan_run.sh "$out/${tag}_tsan.log" 40 > "$out/tsan.out" 2>&1
python tools/tsan_summary.py "$out/${tag}_tsan.log" > "$out/${tag}_tsan_summary.txt" 2>&1
rm -f "$out/${tag}_tsan.log"   # 600 KB of reports about the HIP runtime's own threads: the summary travels
python bench.py --no-extras --pipeline 1 --cpu-sample 0 --no-latency > "$out/${tag}_bench_one_handle.json" 2> "$out/h1.err"
python bench.py --no-extras --pipeline 3 --cpu-sample 0 --no-latency > "$out/${tag}_bench_three_handles.json" 2> "$out/h3.err"
python bench.py --no-extras --steps 20 --warmup 5 --cpu-sample 0 --no-latency > "$out/${tag}_bench_steps20_warmup5.json" 2> "$out/s20.err"
python bench.py --no-extras --fast-lane-ties --cpu-sample 0 --no-latency > "$out/${tag}_bench_fast_lane_ties.json" 2> "$out/et.err"
python bench.py --no-extras --tail-threshold 0 --in-flight 1 --pipeline 1 --cpu-sample 0 > "$out/${tag}_bench_lockstep_only_pipeline1.json" 2> "$out/ls.err"
python bench.py --no-extras --scene ped6 --batch 4096 --cpu-sample 0 > "$out/${tag}_bench_config1_ped6_b4096.json" 2> "$out/c1.err"
python bench.py --no-extras --scene dyn20 --cpu-sample 0 > "$out/${tag}_bench_config4_dyn20_n100.json" 2> "$out/c4.err"
python bench.py --no-extras --scene dyn20x --cpu-sample 0 > "$out/${tag}_bench_config4_dyn20x_n100.json" 2> "$out/c4x.err"
python bench.py --no-extras --coarse dp --cpu-sample 0 > "$out/${tag}_bench_dp_coarse.json" 2> "$out/dp.err"
python bench.py --no-extras --scene demo80 --coarse dp --cpu-sample 0 > "$out/${tag}_bench_dp_coarse_demo80.json" 2> "$out/dp80.err"
python tests/parity_report.py 8192 > "$out/${tag}_parity_report_8192.json" 2> "$out/pr8.err"
python tests/parity_report.py 4096 --fast-lane-ties --plain > "$out/${tag}_parity_report_4096_fast_lane_ties.json" 2> "$out/pr4.err"
# the raw captures stay on the box: only summaries travel back (gpurun_out/ is capped at 64 MiB)
rm -rf "$out"/kt3 "$out"/kt1 "$out"/pmc_*_FETCH_SIZE "$out"/pmc_*_WRITE_SIZE "$out"/pmc_all_[0-9]
du -sh "$out"; ls -la "$out"; for f in "$out"/*.err; do echo "== $f"; tail -n 3 "$f"; done
