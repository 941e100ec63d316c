mkdir -p gpurun_out/r05h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or team or wave or init_guess or stage_parity or golden" > gpurun_out/r05h/tests.log 2>&1; tail -3 gpurun_out/r05h/tests.log
timeout 600 python tools/bwd_forms_sweep.py 512 1024 2048 3072 4096 6144 8192 > gpurun_out/r05h/bwd_forms.json 2> gpurun_out/r05h/bwd_forms.err; cat gpurun_out/r05h/bwd_forms.err
bash tools/pmc_backward_quick.sh > gpurun_out/r05h/pmcq.log 2>&1; tail -16 gpurun_out/r05h/pmcq.log; cp gpurun_out/pmcq/traffic.json gpurun_out/r05h/traffic.json 2>/dev/null
for t in 3072 4096 6144; do timeout 400 python bench.py --cpu-sample 0 --no-latency --wave-threshold $t > gpurun_out/r05h/bench_w$t.json 2>> gpurun_out/r05h/bench.err; python -c "
import json
d=json.loads(open('gpurun_out/r05h/bench_w$t.json').read().strip().splitlines()[-1]); r=d['roofline']; print('wave<=$t', d['value'], d['one_handle']['value'], d['single_batch']['value'], 'bwd_ms', d['breakdown_ms_per_step']['bwd_ms'], 'frac', r.get('frac'), 'B/ps', r.get('bytes_per_problem_step'), 'full', r.get('frac_full_batch'), r.get('bytes_per_problem_step_full_batch_launch'))"; done
