#!/bin/bash
# GPU probe: falsify or confirm "the B = 32 forward is package-power-bound".  Moves the package power cap
# (rocm-smi --setpoweroverdrive) and, separately, pins the shader clock (--setperfdeterminism), and re-measures ms / W /
# sclk / J of the whole forward and of the three heaviest C-ABI calls under each setting (scripts/power_by_call.py).
# Power-bound  <=>  ms scales like (J - floor * t) / (cap - floor): lowering the cap must lengthen the calls that sit at
# the cap at (nearly) constant joules, and a clock pinned BELOW the power-limited clock must lengthen them ~ 1 / clock at
# LOWER power.  Output: gpurun_out/power_cap_sweep.txt
#     gpurun --timeout 900 -- 'bash scripts/power_cap_sweep.sh'
set -u
mkdir -p gpurun_out
OUT=gpurun_out/power_cap_sweep.txt
: > $OUT
export PROBE_SECS=${PROBE_SECS:-1.5} PROBE_CALLS=lh_intra_block,lh_inter_block,lh_local_attn,lh_qkv_proj_ln
run() {
    echo "=== setting: $1" | tee -a $OUT
    rocm-smi --showmaxpower --showperflevel 2>&1 | grep -E "Max Graphics|Performance Level|Power" | tee -a $OUT
    timeout 240 python scripts/power_by_call.py 2>&1 | grep -E "ms .* W|sum over" | tee -a $OUT
}
run "default"
for W in ${CAPS:-1200 1000 800}; do
    echo "--- rocm-smi --setpoweroverdrive $W" | tee -a $OUT
    rocm-smi --setpoweroverdrive $W --autorespond y 2>&1 | grep -v "^$\|====" | head -6 | tee -a $OUT
    run "power cap $W W"
done
rocm-smi --resetpoweroverdrive --autorespond y 2>&1 | grep -v "^$\|====" | head -4 | tee -a $OUT
for F in ${CLOCKS:-2000 1700 1400}; do
    echo "--- rocm-smi --setperfdeterminism $F" | tee -a $OUT
    rocm-smi --setperfdeterminism $F 2>&1 | grep -v "^$\|====" | head -6 | tee -a $OUT
    run "perf determinism sclk <= $F MHz"
done
rocm-smi --resetperfdeterminism 2>&1 | grep -v "^$\|====" | head -4 | tee -a $OUT
run "default again"
