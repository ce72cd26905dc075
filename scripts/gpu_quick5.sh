mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not embedder" > gpurun_out/pytest_gpu_sep.log 2>&1; tail -3 gpurun_out/pytest_gpu_sep.log
for b in 32 256; do
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch $b > gpurun_out/bench_q5_b$b.json 2>> gpurun_out/bench.err
python - gpurun_out/bench_q5_b$b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(' ms/step %.3f  frames/s %.0f' % (d['ms_per_step'], d['value']), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
done
