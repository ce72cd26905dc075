TUNES="_ 10=1 11=1 12=1 10=1,11=1,12=1" REPS=3 bash scripts/gpu.sh tunes 2>&1 | cut -c1-300
