set -x
mkdir -p gpurun_out
R=$PWD
timeout 120 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 500 python -m pytest tests -m gpu -q -x -k embedder > gpurun_out/pytest_gpu_emb.log 2>&1; tail -5 gpurun_out/pytest_gpu_emb.log
timeout 300 python bench.py --mode embed --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_embed_b64.json 2> gpurun_out/bench_embed.err; cat gpurun_out/bench_embed_b64.json
tail -5 gpurun_out/bench_embed.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o r1 -- python $R/bench.py --mode embed --steps 1 --warmup 1 --batch 16 --no-cpu-baseline > $R/gpurun_out/prof_embed.log 2>&1; echo "rocprof rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_e/r1_results.db $R/gpurun_out/embed_kernel_stats.csv
head -24 $R/gpurun_out/embed_kernel_stats.csv
