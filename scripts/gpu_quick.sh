set -x
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
B="timeout 200 python bench.py --no-cpu-baseline"
$B --steps 5 --warmup 2 $BENCH_ARGS > gpurun_out/bench_a.json 2>> gpurun_out/bench.err
$B --mode stream --steps 625 --warmup 50 > gpurun_out/bench_stream.json 2>> gpurun_out/bench.err
tail -5 gpurun_out/bench.err
cat gpurun_out/bench_stream.json
python - gpurun_out/bench_a.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(' ms/step %.3f  frames/s %.0f  rtf %.2e  roof %s %.3f' % (d['ms_per_step'], d['value'], d['rtf'], d['roofline']['kernel'], d['roofline']['frac']))
print('  ', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
