#!/bin/bash
# Samples rocm-smi (engine clock, package power, temperatures) while tools/soak.py keeps the C2 pipeline busy.
# Runs on the GPU box: gpurun -- bash tools/power_probe.sh ; outputs under gpurun_out/.
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/soak.py 1500 > gpurun_out/power_soak.json 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" 
  echo ---
  sleep 1
done > gpurun_out/power_probe.log 2>&1
wait $PID
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" > gpurun_out/power_idle.log
rocm-smi --showmaxpower 2>/dev/null | grep -i power >> gpurun_out/power_idle.log
