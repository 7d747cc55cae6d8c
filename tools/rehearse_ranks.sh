#!/bin/bash
# N ranks of bench.py on ONE GPU with gloo collectives (rehearsal of the multi-GPU job's control flow)
n=${1:-4}; shift
if [ "$n" -gt 5 ]; then echo "at most 5 ranks on one GPU box (process guard: 6 processes)"; exit 2; fi
IPCR_BENCH_ONE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus $n --steps 60 --warmup 5 --backend gloo --records 4 --record-len 50000000 "$@" > gpurun_out/rh.json 2> gpurun_out/rh.err
echo "rc=$? $(grep -h 'products, the set-up' gpurun_out/rh.err | head -2) $(cut -c1-100 gpurun_out/rh.json | head -1)"
