#!/bin/bash
# Sample clocks / power with rocm-smi while the default bench runs (diagnostic: is the step power- or clock-capped?).
cd $GRAFT_REPO_ROOT
python bench.py --steps 12000 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
for i in $(seq 1 45); do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|Power" | tr '\n' ' '
  echo
  kill -0 $BP 2>/dev/null || break
  sleep 1
done
wait $BP
tail -c 200 gpurun_out/power_bench.json
