#!/bin/bash
# round 3, batch O (GPU box): does the slab copy's speed depend on which socket the file-reading threads run on?
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
f=/tmp/ipcr_m.fa
python3 tools/fasta_load.py make $f || exit 1
cat $f > /dev/null
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) $(cat $d/local_cpulist 2>/dev/null)"; done
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; nproc
for cpus in 0-63,128-191 64-127,192-255 0-15 64-79 ""; do
  echo "== cpus '$cpus'"
  if [ -n "$cpus" ]; then pre="taskset -c $cpus"; else pre=""; fi
  IPCR_DEBUG_TIMES=2 $pre timeout -k 10 120 tools/ubench/fasta_load $f 3 > /tmp/o.txt 2>&1; grep "native load\|copy of slab  [7-9]\|rror" /tmp/o.txt | tail -5
done
