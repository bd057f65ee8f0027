#!/bin/bash
# shader clock while the Gram kernel runs back to back (stage-3 shape), sampled with rocm-smi from a second process
python3 scripts/profile_fit.py --periods 381 --n 24963 --bw 20 --reps 4000 > gpurun_out/clock_load.log 2>&1 &
PID=$!
sleep 9
for i in 1 2 3 4 5; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | tr -s ' ' | head -4; echo --; sleep 0.5; done
wait $PID
tail -1 gpurun_out/clock_load.log
