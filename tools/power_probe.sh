#!/bin/bash
# power_probe.sh — socket power / clock while a kernel runs in a loop: "$@" is started in the background and
# rocm-smi is sampled once a second for N seconds (env N, default 8).  Shows whether a kernel sits at the power cap.
N=${N:-8}
"$@" > /dev/null 2>&1 &
pid=$!
sleep 2
for i in $(seq 1 $N); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr -s ' ' | paste -sd' ' | sed -e "s/=*//g" -e "s/GPU\[0\] : //g" | cut -c1-220
    sleep 1
done
kill $pid 2>/dev/null
wait $pid 2>/dev/null
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
