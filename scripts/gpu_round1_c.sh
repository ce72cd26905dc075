set -x
mkdir -p gpurun_out
R=$PWD
timeout 600 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
B="timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
$B > gpurun_out/bench_f16x3_mt1.json 2>> gpurun_out/bench.err
$B --tune 0=2 > gpurun_out/bench_f16x3_mt2.json 2>> gpurun_out/bench.err
$B --gemm f32 > gpurun_out/bench_f32.json 2>> gpurun_out/bench.err
$B --batch 1 > gpurun_out/bench_b1.json 2>> gpurun_out/bench.err
$B --batch 256 --steps 2 --warmup 1 > gpurun_out/bench_b256.json 2>> gpurun_out/bench.err
for f in gpurun_out/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(' ms/step %.3f  frames/s %.0f  rtf %.2e  roof %s %.3f' % (d['ms_per_step'], d['value'], d['rtf'], d['roofline']['kernel'], d['roofline']['frac']))
    print('  ', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})
except Exception as e: print('ERR', e)
PY
done
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1; echo "rocprof write rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/prof_sq -o r1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_sq.log 2>&1; echo "rocprof sq rc=$?"
ls -la $R/gpurun_out/prof_*/
tail -3 $R/gpurun_out/bench.err
