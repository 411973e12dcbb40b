#!/bin/bash
# sample power / clocks while a kernel loop runs:  tools/power_probe.sh "<python command>"
( eval "$1" > gpurun_out/power_cmd.log 2>&1 ) &
PID=$!
sleep ${2:-25}
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | sed 's/^/  /'
  echo ---
  sleep 1
done
wait $PID
tail -2 gpurun_out/power_cmd.log
