set -x
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof rc=$?"
ls -R $R/gpurun_out/prof_stats | head -30
find $R/gpurun_out/prof_stats -name "*kernel_stats*" | head -2 | xargs -r head -25
