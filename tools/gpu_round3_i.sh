#!/bin/bash
# round 3, batch I (GPU box): the multi-GPU control flow of bench.py on one GPU -- one-rank RCCL job (native exchange), two gloo ranks
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03i
mkdir -p $out
IPCR_EXCHANGE_SELFTEST=1 timeout -k 10 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/selftest.json 2> $out/selftest.err || { echo "selftest failed"; tail -20 $out/selftest.err; exit 1; }
python3 -c "import json; d=json.loads(open('$out/selftest.json').read().splitlines()[-1]); c=d['config']; print('selftest', d['value'], d['ms_per_step'], 'rccl_ranks', c['rccl_ranks'], 'native', c['native_exchange'], 'device_path', c['device_path'], c['parallelism'][:80])"
IPCR_BENCH_ONE_DEVICE=1 timeout -k 10 900 python3 bench.py --gpus 2 --backend gloo --steps 100 --warmup 10 --no-cpu-baseline > $out/gloo2.json 2> $out/gloo2.err || { echo "gloo 2-rank failed"; tail -20 $out/gloo2.err; exit 1; }
python3 -c "import json; d=json.loads(open('$out/gloo2.json').read().splitlines()[-1]); c=d['config']; print('gloo2', d['n_gpus'], d['value'], d['ms_per_step'], 'native', c['native_exchange'], {k: v.get('gbases_per_s') for k, v in c['other_workloads'].items()})"
