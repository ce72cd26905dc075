# GPU parity suite + bench (batch 32, batch 1, streaming) after a kernel change
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for a in "--batch 32" "--batch 1" "--mode stream"; do
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $a > gpurun_out/bench_chk.json 2>> gpurun_out/bench.err
python - gpurun_out/bench_chk.json "$a" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'ms/step %.3f' % d['ms_per_step'], d['value'], d['unit'], {k: round(v,3) for k,v in d.get('kernels_ms_per_step', {}).items()})
PY
done
