# cache-level counters for the attention / QKV kernels (separate --pmc passes, kernel trace only)
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${TUNE}"
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_c$i -o r1 -- $P > $R/gpurun_out/prof_c$i.log 2>&1; echo "rocprof pass $i rc=$?"
  python $R/scripts/rocpd_summary.py /tmp/prof_c$i/r1_results.db $R/gpurun_out/pmc_c$i.csv --pmc
  grep "k_local_attn\|k_qkv\|k_proj_ln" $R/gpurun_out/pmc_c$i.csv
done
