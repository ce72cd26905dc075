# A/B of two builds of the library inside one box: LOOKONCE_HIP_LIB selects the build (timing only: the older build
# may not match the current weight packing)
mkdir -p gpurun_out
for rep in 1 2; do
for lib in ${LIBS:-_lookonce_hip_prev.so _lookonce_hip.so}; do
for b in 32; do
LOOKONCE_HIP_LIB=$PWD/lookoncetohear_amd/$lib timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --batch $b > gpurun_out/bench_ab.json 2>> gpurun_out/bench.err
python - gpurun_out/bench_ab.json $lib $b <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-26s B %s  ms/step %.3f' % (sys.argv[2], sys.argv[3], d['ms_per_step']), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
done; done; done
