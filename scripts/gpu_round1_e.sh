set -x
mkdir -p gpurun_out
R=$PWD
timeout 300 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print(' ms/step %.3f  frames/s %.0f  rtf %.2e  roof %s %.3f' % (d['ms_per_step'], d['value'], d['rtf'], d['roofline']['kernel'], d['roofline']['frac']))
print('  ', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()}); print(d['cpu_baseline'])
PY
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r1 -- $P > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_stats/r1_results.db $R/gpurun_out/kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_sq -o r1 -- $P > $R/gpurun_out/prof_sq.log 2>&1; echo "rocprof sq rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_sq/r1_results.db $R/gpurun_out/pmc_sq.csv --pmc
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/prof_sq2 -o r1 -- $P > $R/gpurun_out/prof_sq2.log 2>&1; echo "rocprof sq2 rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_sq2/r1_results.db $R/gpurun_out/pmc_sq2.csv --pmc
head -12 $R/gpurun_out/kernel_stats.csv
