# Collects the round's committed evidence: bench line, rocprofv3 kernel stats, PMC passes (separate runs, as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass), all summarised on the box.
set -x
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r1 -- $P > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_stats/r1_results.db $R/gpurun_out/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_$C -o r1 -- $P > $R/gpurun_out/prof_$C.log 2>&1; echo "rocprof $C rc=$?"
  python $R/scripts/rocpd_summary.py /tmp/prof_$C/r1_results.db $R/gpurun_out/pmc_$C.csv --pmc
done
python $R/scripts/make_pmc_traffic.py $R/gpurun_out/pmc_FETCH_SIZE.csv $R/gpurun_out/pmc_WRITE_SIZE.csv 32 $R/gpurun_out/pmc_traffic.json
mkdir -p $R/profiles && cp $R/gpurun_out/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_sq -o r1 -- $P > $R/gpurun_out/prof_sq.log 2>&1; echo "rocprof sq rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_sq/r1_results.db $R/gpurun_out/pmc_sq.csv --pmc
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/prof_sq2 -o r1 -- $P > $R/gpurun_out/prof_sq2.log 2>&1; echo "rocprof sq2 rc=$?"
python $R/scripts/rocpd_summary.py /tmp/prof_sq2/r1_results.db $R/gpurun_out/pmc_sq2.csv --pmc
cd $R
# the bench line last, so that it picks up the freshly measured traffic (profiles/pmc_traffic.json)
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
cat gpurun_out/kernel_stats.csv | head -40; cat gpurun_out/pmc_traffic.json | head -60; cat gpurun_out/bench.json | cut -c1-1500
