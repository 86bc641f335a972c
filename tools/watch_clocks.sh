#!/bin/bash
# sample power / clocks while the hot kernel runs back to back
cd "${GRAFT_REPO_ROOT:-/root/repo}"
(python bench.py --no-cpu --no-config5 --no-host-call --warmup 5 --steps 3000 > gpurun_out/watch_bench.json 2>/dev/null) &
BP=$!
for i in $(seq 1 80); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | awk '{printf "%s ", $0} END {print ""}' | sed 's/GPU\[\([0-9]\)\]\s*:/g\1/g; s/clock level//g; s/Current Socket Graphics Package Power (W)/W/g'
  kill -0 $BP 2>/dev/null || break
  sleep 0.2
done | awk '{print NR": "$0}' | grep -v "^$" | tail -45
wait $BP
python -c "
import json; d=json.loads(open('gpurun_out/watch_bench.json').read()); print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'])"
