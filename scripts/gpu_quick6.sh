mkdir -p gpurun_out
for d in 0 1 2 3 4 6; do
LH_ATTN_DBG=$d timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_q6.json 2>> gpurun_out/bench.err
python - gpurun_out/bench_q6.json $d <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('dbg', sys.argv[2], ' attn ms/step %.3f' % d['kernels_ms_per_step']['lh_local_attn'])
PY
done
