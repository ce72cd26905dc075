mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_st -o r1 -- python $R/bench.py --mode stream --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof_stream.log 2>&1; echo "rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_st/r1_results.db $R/gpurun_out/stream_kernel_stats.csv
head -40 $R/gpurun_out/stream_kernel_stats.csv
tail -2 $R/gpurun_out/prof_stream.log | cut -c1-400
