#!/bin/bash
# Samples socket power and shader clock (rocm-smi) while the bench's denoise loop runs: is the step power-capped?
cd /root/repo
python bench.py --steps 700 --warmup 5 --no-cpu-baseline > /tmp/bench_pw.log 2>&1 &
BP=$!
sleep 22
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' | cut -c1-400; echo
  sleep 1
done
wait $BP
tail -1 /tmp/bench_pw.log | cut -c1-200
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
