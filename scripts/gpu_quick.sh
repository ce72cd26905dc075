# quick timing of the B=32 bench (per-call breakdown), optional LOOKONCE_HIP_LIB variants in $LIBS
mkdir -p gpurun_out
for rep in 1 2; do
for lib in ${LIBS:-_lookonce_hip.so}; do
LOOKONCE_HIP_LIB=$PWD/lookoncetohear_amd/$lib timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ${ARGS} > gpurun_out/bench_q.json 2>> gpurun_out/bench.err
python - gpurun_out/bench_q.json $lib <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-26s ms/step %.3f' % (sys.argv[2], d['ms_per_step']), {k: round(v,3) for k,v in d.get('kernels_ms_per_step', {}).items()}, d.get('metric_sums', [0])[:1])
PY
done; done
