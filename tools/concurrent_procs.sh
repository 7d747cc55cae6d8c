#!/bin/bash
# N independent bench processes on one GPU at once (time-sliced queues): every pass of every process is checked
n=${1:-4}; shift
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps 300 --warmup 5 --records 4 --record-len 50000000 "$@" > gpurun_out/cp_$i.json 2> gpurun_out/cp_$i.err &
done
wait
for i in $(seq 1 $n); do echo "proc $i: $(grep -h 'products, the set-up' gpurun_out/cp_$i.json gpurun_out/cp_$i.err | head -1) $(cut -c1-90 gpurun_out/cp_$i.json | head -1)"; done
