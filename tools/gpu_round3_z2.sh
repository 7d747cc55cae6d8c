#!/bin/bash
# round 3, batch Z2 (GPU box): after the stack drain -- C4 profile, the index kernel's cost map over panels, the chunk path
# under a 1024-row panel, two-rank rehearsal of the multi-GPU line
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r03z
mkdir -p $out
bash tools/profile_round.sh r03 prof c4 || exit 1
bash tools/gpu_round3_x.sh > $out/costmap.txt 2>&1; cat $out/costmap.txt
for cb in 4000000 32000000; do
  CHUNK_PANEL_ROWS=1024 GPU_MAX_HW_QUEUES=4 timeout -k 10 200 ipcr_amd/chunk_workers 125000000 $cb 1 8 16 > $out/chunk_c4_$cb.json 2> $out/chunk_c4_$cb.err || { tail -3 $out/chunk_c4_$cb.err; }
  cut -c1-900 $out/chunk_c4_$cb.json
done
IPCR_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 > $out/bench2.json 2> $out/bench2.err || { echo "bench2 failed"; tail -15 $out/bench2.err; exit 1; }
python3 -c "import json; d=json.loads(open('$out/bench2.json').read().splitlines()[-1]); print('2 ranks:', d['value'], d['n_gpus'], json.dumps(d['config']['other_workloads'])[:900])"
