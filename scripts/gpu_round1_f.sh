set -x
mkdir -p gpurun_out
R=$PWD
timeout 300 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
B="timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
$B > gpurun_out/bench_pipe.json 2>> gpurun_out/bench.err
$B --tune 2=1 > gpurun_out/bench_nopipe.json 2>> gpurun_out/bench.err
$B --batch 1 --steps 10 > gpurun_out/bench_b1.json 2>> gpurun_out/bench.err
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
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r1 -- $P > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_stats/r1_results.db $R/gpurun_out/kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_sq -o r1 -- $P > $R/gpurun_out/prof_sq.log 2>&1; echo "rocprof sq rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_sq/r1_results.db $R/gpurun_out/pmc_sq.csv --pmc
head -12 $R/gpurun_out/kernel_stats.csv
