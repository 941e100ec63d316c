mkdir -p gpurun_out/r05g
python -m pytest tests -m gpu -x -q > gpurun_out/r05g/gputest.log 2>&1; tail -4 gpurun_out/r05g/gputest.log
for k in 1 2; do timeout 400 python bench.py --cpu-sample 0 --no-latency > gpurun_out/r05g/bench_$k.json 2>> gpurun_out/r05g/bench.err; python -c "
import json
d=json.loads(open('gpurun_out/r05g/bench_$k.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['one_handle']['value'], d['single_batch']['value'], d['breakdown_ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r05g/kt1 -- python bench.py --steps 6 --warmup 2 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency > gpurun_out/r05g/bench_p1.json 2> gpurun_out/r05g/kt1.err
{ python tools/prof_summary.py gpurun_out/r05g/kt1 --iters 1,2,5,10,20,40,80; python tools/phase_summary.py gpurun_out/r05g/kt1; } > gpurun_out/r05g/kernel_stats.txt 2>&1
rm -rf gpurun_out/r05g/kt1
head -24 gpurun_out/r05g/kernel_stats.txt
