mkdir -p gpurun_out
for t in "3=1" "3=0" "3=1" "3=0"; do
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --tune $t > gpurun_out/bench_q7.json 2>> gpurun_out/bench.err
python - gpurun_out/bench_q7.json $t <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], ' ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if 'block' in k})
PY
done
