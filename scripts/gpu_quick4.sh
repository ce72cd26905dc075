set -x
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
B="timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
LOOKONCE_LSTM_WAVES=8 $B > gpurun_out/bench_w8.json 2>> gpurun_out/bench.err
LOOKONCE_LSTM_WAVES=4 $B > gpurun_out/bench_w4.json 2>> gpurun_out/bench.err
LOOKONCE_LSTM_WAVES=8 $B --batch 256 --steps 2 --warmup 1 > gpurun_out/bench_w8_b256.json 2>> gpurun_out/bench.err
LOOKONCE_LSTM_WAVES=8 $B --batch 1 --steps 10 > gpurun_out/bench_w8_b1.json 2>> gpurun_out/bench.err
LOOKONCE_LSTM_WAVES=8 timeout 200 python bench.py --no-cpu-baseline --mode stream --steps 625 --warmup 50 > gpurun_out/bench_stream.json 2>> gpurun_out/bench.err
for f in gpurun_out/bench_w8.json gpurun_out/bench_w4.json gpurun_out/bench_w8_b256.json gpurun_out/bench_w8_b1.json; do echo $f; python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(' ms/step %.3f  frames/s %.0f  rtf %.2e  roof %s %.3f' % (d['ms_per_step'], d['value'], d['rtf'], d['roofline']['kernel'], d['roofline']['frac']))
print('  ', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
done
cut -c1-330 gpurun_out/bench_stream.json
