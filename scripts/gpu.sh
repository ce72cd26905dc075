#!/bin/bash
# One entry point for everything that runs on the MI355X box through gpurun:
#     gpurun --timeout 900 -- 'bash scripts/gpu.sh <task> [args]'
# Tasks write their results under gpurun_out/ (merged back by gpurun); the summaries worth keeping are copied into
# profiles/ by hand afterwards.
set -u
R=$PWD
mkdir -p gpurun_out
task=${1:-check}; shift || true

bench_line() {      # bench_line <json file> <label>: one human-readable line from a bench.py JSON line
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%-30s ms/step %.3f  %.4g %s' % (sys.argv[2], d['ms_per_step'], d['value'], d['unit']),
          {k: round(v, 3) for k, v in d.get('kernels_ms_per_step', {}).items()})
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}

case $task in
ubench)             # per-SIMD issue model micro-benchmark (scripts/ubench/gen_issue_model.py; binary built on the CPU side)
    timeout 300 scripts/ubench/issue_model > gpurun_out/ubench_issue_model.txt 2>&1; echo "rc=$?"
    cat gpurun_out/ubench_issue_model.txt ;;
check)              # GPU parity suite + bench lines (batch 32, batch 1, streaming)
    timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
    for a in "--batch 32" "--batch 1" "--mode stream"; do
        timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-power $a > gpurun_out/bench_chk.json 2>> gpurun_out/bench.err
        bench_line gpurun_out/bench_chk.json "$a"
    done ;;
ab)                 # same-box A/B of library builds: LIBS="_lookonce_hip_x.so _lookonce_hip.so" BATCH=32 REPS=2
    for rep in $(seq ${REPS:-2}); do for lib in ${LIBS:-_lookonce_hip.so}; do for b in ${BATCH:-32}; do
        LOOKONCE_HIP_LIB=$R/lookoncetohear_amd/$lib timeout 200 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-secondary --no-power --batch $b ${BENCH_ARGS:-} > gpurun_out/bench_ab.json 2>> gpurun_out/bench.err
        bench_line gpurun_out/bench_ab.json "$lib B=$b"
    done; done; done ;;
tunes)              # same-box A/B of lh_set_tuning switches: TUNES="_ 5=2 8=0,9=0" (_ = defaults) BATCH=32 REPS=2
    for rep in $(seq ${REPS:-2}); do for t in ${TUNES:-_}; do for b in ${BATCH:-32}; do
        ta=""; [ "$t" != "_" ] && ta="--tune $t"
        timeout 200 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-secondary --no-power --batch $b $ta > gpurun_out/bench_ab.json 2>> gpurun_out/bench.err
        bench_line gpurun_out/bench_ab.json "tune $t B=$b"
    done; done; done ;;
lab)                # recurrent-kernel lab: LIBS="_lookonce_hip.so _lookonce_hip_x.so" TUNES="_ 2=2 5=2 9=0" (scripts/lab_recur.py)
    for lib in ${LIBS:-_lookonce_hip.so}; do
        LOOKONCE_HIP_LIB=$R/lookoncetohear_amd/$lib timeout 300 python scripts/lab_recur.py --tunes "${TUNES:-_ 2=2,5=2}" ${LAB_ARGS:-} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lab_recur.txt
    done ;;
bench)              # the default bench line, as the driver runs it
    timeout 900 python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
    cut -c1-3000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err ;;
profile)            # rocprofv3 kernel stats + PMC passes (separate runs) of the B=32 bench; then the bench line itself
    cd /tmp && export TMPDIR=/tmp
    P="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-power"
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r1 -- $P > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof rc=$?"
    python $R/scripts/rocpd_summary.py /tmp/prof_stats/r1_results.db $R/gpurun_out/kernel_stats.csv
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_$C -o r1 -- $P > $R/gpurun_out/prof_$C.log 2>&1; echo "rocprof $C rc=$?"
        python $R/scripts/rocpd_summary.py /tmp/prof_$C/r1_results.db $R/gpurun_out/pmc_$C.csv --pmc
    done
    [ "${LITE:-0}" = "1" ] && { python $R/scripts/make_pmc_traffic.py $R/gpurun_out/pmc_FETCH_SIZE.csv $R/gpurun_out/pmc_WRITE_SIZE.csv 32 $R/gpurun_out/pmc_traffic.json; cd $R; head -12 gpurun_out/kernel_stats.csv; exit 0; }      # LITE=1: stats + traffic passes only
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_sq -o r1 -- $P > $R/gpurun_out/prof_sq.log 2>&1; echo "rocprof sq rc=$?"
    python $R/scripts/rocpd_summary.py /tmp/prof_sq/r1_results.db $R/gpurun_out/pmc_sq.csv --pmc
    timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/prof_sq2 -o r1 -- $P > $R/gpurun_out/prof_sq2.log 2>&1; echo "rocprof sq2 rc=$?"
    python $R/scripts/rocpd_summary.py /tmp/prof_sq2/r1_results.db $R/gpurun_out/pmc_sq2.csv --pmc
    python $R/scripts/make_pmc_traffic.py $R/gpurun_out/pmc_FETCH_SIZE.csv $R/gpurun_out/pmc_WRITE_SIZE.csv 32 $R/gpurun_out/pmc_traffic.json $R/gpurun_out/pmc_sq.csv $R/gpurun_out/pmc_sq2.csv $R/gpurun_out/kernel_stats.csv
    cd $R
    head -40 gpurun_out/kernel_stats.csv; head -60 gpurun_out/pmc_traffic.json ;;
profile_embed)      # the same evidence set for BASELINE configs[4]: kernel stats + FETCH / WRITE / sq / sq2 passes
    cd /tmp && export TMPDIR=/tmp       # of `bench.py --mode embed`; merges an `embed` section into gpurun_out/pmc_traffic.json
    export LOOKONCE_PACK_ON_HOST=1     # the device-side weight packers' ~6000 tiny indexing launches crash rocprofv3 --pmc
    export LOOKONCE_EMB_STREAMS=1      # one stream: every launch is a whole-batch launch (5 forwards = 5 calls of each stage)
    P="python $R/bench.py --mode embed --steps 3 --warmup 1 --no-cpu-baseline"
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/eprof_stats -o r1 -- $P > $R/gpurun_out/eprof_stats.log 2>&1; echo "rocprof rc=$?"
    python $R/scripts/rocpd_summary.py /tmp/eprof_stats/r1_results.db $R/gpurun_out/embed_kernel_stats.csv
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/eprof_$C -o r1 -- $P > $R/gpurun_out/eprof_$C.log 2>&1; echo "rocprof $C rc=$?"
        python $R/scripts/rocpd_summary.py /tmp/eprof_$C/r1_results.db $R/gpurun_out/embed_pmc_$C.csv --pmc
    done
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/eprof_sq -o r1 -- $P > $R/gpurun_out/eprof_sq.log 2>&1; echo "rocprof sq rc=$?"
    python $R/scripts/rocpd_summary.py /tmp/eprof_sq/r1_results.db $R/gpurun_out/embed_pmc_sq.csv --pmc
    timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/eprof_sq2 -o r1 -- $P > $R/gpurun_out/eprof_sq2.log 2>&1; echo "rocprof sq2 rc=$?"
    python $R/scripts/rocpd_summary.py /tmp/eprof_sq2/r1_results.db $R/gpurun_out/embed_pmc_sq2.csv --pmc
    [ -f $R/gpurun_out/pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $R/gpurun_out/pmc_traffic.json
    python $R/scripts/make_pmc_traffic_embed.py $R/gpurun_out/embed_kernel_stats.csv $R/gpurun_out/embed_pmc_FETCH_SIZE.csv $R/gpurun_out/embed_pmc_WRITE_SIZE.csv 64 5 $R/gpurun_out/pmc_traffic.json $R/gpurun_out/embed_pmc_sq.csv $R/gpurun_out/embed_pmc_sq2.csv
    cd $R; head -24 gpurun_out/embed_kernel_stats.csv ;;
round_end)          # one call for a small change late in a round: same-box A/B against LIBS' first entry, the default bench line,
                    # kernel stats + the dispatch timeline of one step, then the GPU suite and smoke (most valuable first)
    for rep in 1 2; do for lib in ${LIBS:-_lookonce_hip_old.so _lookonce_hip.so}; do
        LOOKONCE_HIP_LIB=$R/lookoncetohear_amd/$lib timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-power --no-gpu-library-baseline > gpurun_out/bench_ab.json 2>> gpurun_out/bench.err
        bench_line gpurun_out/bench_ab.json "$lib B=32"
    done; done | tee gpurun_out/ab_round_end.txt
    timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
    cut -c1-600 gpurun_out/bench.json
    ( cd /tmp && export TMPDIR=/tmp
      timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-power --no-gpu-library-baseline > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof rc=$?"
      python $R/scripts/rocpd_summary.py /tmp/prof_stats/r1_results.db $R/gpurun_out/kernel_stats.csv
      python $R/scripts/rocpd_summary.py /tmp/prof_stats/r1_results.db $R/gpurun_out/step_timeline.txt --timeline )
    tail -3 gpurun_out/step_timeline.txt
    timeout ${SUITE_TIMEOUT:-560} python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.txt
    timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee -a gpurun_out/pytest_gpu.txt ;;
timelines)          # dispatch timelines of one step (scripts/rocpd_summary.py --timeline): the B=32 forward with the default kernels and
                    # with TUNES' switches, one embedder forward on one stream; then the default bench line
    cd /tmp && export TMPDIR=/tmp
    for t in ${TUNES:-_ 5=2}; do
        ta=""; [ "$t" != "_" ] && ta="--tune $t"
        timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tl_$t -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-power --no-gpu-library-baseline $ta > $R/gpurun_out/tl_$t.log 2>&1; echo "rocprof $t rc=$?"
        python $R/scripts/rocpd_summary.py /tmp/tl_$t/r1_results.db $R/gpurun_out/step_timeline_$t.txt --timeline
    done
    LOOKONCE_PACK_ON_HOST=1 LOOKONCE_EMB_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tl_embed -o r1 -- python $R/bench.py --mode embed --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/tl_embed.log 2>&1; echo "rocprof embed rc=$?"
    python $R/scripts/rocpd_summary.py /tmp/tl_embed/r1_results.db $R/gpurun_out/step_timeline_embed.txt --timeline k_emb_std
    cd $R
    timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
    cut -c1-400 gpurun_out/bench.json; tail -3 gpurun_out/step_timeline_*.txt ;;
sweep)              # the separator over the batch size, whole-clip launches against the default time windows (scripts/batch_sweep.py)
    for cfg in "1 1" "0 0"; do set -- $cfg; echo "time_chunks $1 time_chunks_small $2"
        LOOKONCE_TIME_CHUNKS=$1 LOOKONCE_TIME_CHUNKS_SMALL=$2 python scripts/batch_sweep.py ${SWEEP_B:-} 2>&1 | grep "B ="
    done | tee gpurun_out/batch_sweep.txt ;;
chunks)             # Net.time_chunks: the bit-identity test, then same-box A/B of the default line over K (REPS x K in "1 2 3 4")
    timeout 600 python -m pytest tests/test_gpu_modes.py -m gpu -x -q -k "time_chunks" 2>&1 | tail -5 | tee gpurun_out/pytest_chunks.txt
    for rep in $(seq ${REPS:-2}); do for k in ${KS:-1 2 3 4}; do
        timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-power --no-gpu-library-baseline --time-chunks $k ${BENCH_ARGS:-} > gpurun_out/bench_ab.json 2>> gpurun_out/bench.err
        bench_line gpurun_out/bench_ab.json "time_chunks $k B=32"
    done; done | tee gpurun_out/ab_time_chunks.txt ;;
final)              # A/B of bench.py's event bracket (every call / every 4th call of the dominant entry point), the default line, the
                    # embedder line, the GPU suite, smoke
    for rep in 1 2; do for es in 1 4; do
        LOOKONCE_BENCH_EVENT_STRIDE=$es timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-power --no-gpu-library-baseline > gpurun_out/bench_ab.json 2>> gpurun_out/bench.err
        bench_line gpurun_out/bench_ab.json "event stride $es B=32"
    done; done | tee gpurun_out/ab_event_stride.txt
    timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
    cut -c1-400 gpurun_out/bench.json
    timeout 300 python bench.py --mode embed > gpurun_out/bench_embed.json 2>> gpurun_out/bench.err; echo "embed rc=$?"
    cut -c1-300 gpurun_out/bench_embed.json
    timeout ${SUITE_TIMEOUT:-420} python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.txt
    timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee -a gpurun_out/pytest_gpu.txt ;;
*)  echo "unknown task $task"; exit 2 ;;
esac
