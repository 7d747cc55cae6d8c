#!/bin/bash
# sample power / clocks while the bench loop runs (is the sustained sweep clock- or power-limited?)
python bench.py --no-cpu-baseline --steps 20000 --warmup 5 "$@" > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "GPU\[0\].*(sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory))" | sed 's/\s\+/ /g'
  echo --
  sleep 0.5
done
wait $pid
python tools/bench_line.py < gpurun_out/clock_bench.json
