#!/bin/bash
# round 3, batch J (GPU box): zero-copy experiment of the chunk path, then the default bench line with in-run traffic (timed)
set -o pipefail
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/r03j
mkdir -p $out
KNOBS=$'A=0\nIPCR_CHUNK_ZEROCOPY=1' WORKERS="8 16" bash tools/gpu_round3_e.sh || exit 1
s=$(date +%s)
timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -15 $out/bench.err; exit 1; }
e=$(date +%s)
python3 -c "import json; d=json.loads(open('$out/bench.json').read().splitlines()[-1]); r=d['roofline']; print('default bench took', $e-$s, 's; value', d['value'], 'frac', r['frac'], 'traffic', r['traffic'], r.get('traffic_over_algorithmic'), r.get('traffic_source','')[:60])"
