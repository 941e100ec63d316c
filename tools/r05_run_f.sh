mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "four_row or speculative or tail_kernel_is_bit or exit_paths_at_batch or size_independent or full_batch or 65536" > gpurun_out/r05f/tests.log 2>&1; tail -6 gpurun_out/r05f/tests.log
b() { name=$1; shift; timeout 400 python bench.py --cpu-sample 0 --no-latency "$@" > gpurun_out/r05f/bench_$name.json 2>> gpurun_out/r05f/bench.err; python -c "
import json
d=json.loads(open('gpurun_out/r05f/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', d['value'], d['one_handle']['value'], d['single_batch']['value'], 'bwd_ms', d['breakdown_ms_per_step']['bwd_ms'], 'frac', r.get('frac'), 'bytes', d['device_bytes'])"; }
b default
b wave3072 --wave-threshold 3072
b wave3072_team8192 --wave-threshold 3072 --team-threshold 8192
b wave2048 --wave-threshold 2048
b three_handles --pipeline 3
CILQR_LIB=$PWD/cilqr_amd/lib/variants/libcilqr_hip_bwdilp.so timeout 300 python tools/bwd_forms_sweep.py 1024 2048 4096 8192 16384 65536 > gpurun_out/r05f/bwd_forms_ilp.json 2> gpurun_out/r05f/bwd_forms_ilp.err; cat gpurun_out/r05f/bwd_forms_ilp.err
timeout 300 python tools/bwd_forms_sweep.py 1024 2048 4096 8192 16384 65536 > gpurun_out/r05f/bwd_forms.json 2> gpurun_out/r05f/bwd_forms.err; cat gpurun_out/r05f/bwd_forms.err
