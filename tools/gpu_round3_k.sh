#!/bin/bash
# round 3, batch K (GPU box): chunk rates with the lone-caller zero-copy default, then a two-rank rehearsal of the multi-GPU
# bench line on one card (gloo, both ranks on device 0; the one-process all-devices pool through device slots)
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03k
mkdir -p $out
KNOBS=$'A=0' WORKERS="8 16" bash tools/gpu_round3_e.sh || exit 1
IPCR_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 > $out/bench2.json 2> $out/bench2.err || { echo "bench2 failed"; tail -15 $out/bench2.err; exit 1; }
python3 -c "import json; d=json.loads(open('$out/bench2.json').read().splitlines()[-1]); print('2 ranks:', d['value'], d['n_gpus'], json.dumps(d['config']['other_workloads'].get('scan_chunk_one_process_all_devices'))[:600])"
