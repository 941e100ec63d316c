mkdir -p gpurun_out/r05i
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "speculative or round or exit_paths_at_batch" > gpurun_out/r05i/tests_default.log 2>&1; tail -2 gpurun_out/r05i/tests_default.log
CILQR_ROUND_SCHEDULE=211 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "speculative or round or exit_paths_at_batch or full_size or tiled" > gpurun_out/r05i/tests_211.log 2>&1; tail -2 gpurun_out/r05i/tests_211.log
b() { name=$1; shift; timeout 400 python bench.py --cpu-sample 0 --no-latency "$@" > gpurun_out/r05i/bench_$name.json 2>> gpurun_out/r05i/bench.err; python -c "
import json
d=json.loads(open('gpurun_out/r05i/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['one_handle']['value'] if d.get('one_handle') else None, d['single_batch']['value'] if d.get('single_batch') else None, d['breakdown_ms_per_step'])"; }
b s22_a
CILQR_ROUND_SCHEDULE=211 b s211_a
b s22_b
CILQR_ROUND_SCHEDULE=211 b s211_b
CILQR_BENCH_FORCE_DIST=1 python bench.py --cpu-sample 0 --no-latency > gpurun_out/r05i/r05_bench_force_dist.json 2> gpurun_out/r05i/fd.err
CILQR_BENCH_MULTI_DEVICES=0,0 python bench.py --gpus 2 --multi --cpu-sample 0 --no-latency > gpurun_out/r05i/r05_bench_multi_two_shards_one_gpu.json 2> gpurun_out/r05i/mu.err
