R=$PWD
for i in 1 2; do
LOOKONCE_HIP_LIB=$R/lookoncetohear_amd/_lookonce_hip.so timeout 300 python scripts/lab_recur.py --tunes "_ 8=1 8=2 8=3" --which intra --reps 10 2>&1 | grep -v amdgpu.ids
done
TUNES="5=1,9=3,8=1 5=1,9=3,8=2 5=1,9=3,8=3" REPS=3 bash scripts/gpu.sh tunes 2>&1 | cut -c1-75
