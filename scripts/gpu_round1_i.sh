set -x
mkdir -p gpurun_out
R=$PWD
timeout 300 python -m pytest tests -m gpu -q -x -k "render" > gpurun_out/pytest_gpu_render.log 2>&1; tail -5 gpurun_out/pytest_gpu_render.log
timeout 200 python bench.py --mode render --steps 5 --warmup 2 > gpurun_out/bench_render_256.json 2> gpurun_out/bench_render.err; cat gpurun_out/bench_render_256.json
timeout 200 python bench.py --mode render --steps 3 --warmup 1 --rir-len 4096 > gpurun_out/bench_render_4096.json 2>> gpurun_out/bench_render.err; cat gpurun_out/bench_render_4096.json
tail -3 gpurun_out/bench_render.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_r -o r1 -- python $R/bench.py --mode render --steps 3 --warmup 1 --rir-len 4096 --no-cpu-baseline > $R/gpurun_out/prof_render.log 2>&1; echo "rocprof rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_r/r1_results.db $R/gpurun_out/render_kernel_stats.csv
head -8 $R/gpurun_out/render_kernel_stats.csv; tail -8 $R/gpurun_out/render_kernel_stats.csv
