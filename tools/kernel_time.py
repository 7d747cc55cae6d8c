#!/usr/bin/env python3
"""Filter-kernel time of a workload on the resident 3 Gb LCG genome, without result checks (for knob
experiments that change what the kernel does): tools/kernel_time.py [c2|c3] [passes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipcr_amd import engine as E, workloads

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 400
if which == "c3":
    cfg = E.Config(MaxMM=3, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=True)
    pairs = workloads.c3_pairs()
else:
    cfg = E.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    pairs = workloads.c2_pairs()
eng = E.New(cfg)
cp = eng.CompilePanel(pairs)
scs = [eng.NewSimulationScratch(cp), eng.NewSimulationScratch(cp)]
nrec, reclen = 24, 125_000_000
g = E.Genome(nrec * reclen, nrec)
buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
for r in range(nrec):
    E.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed1234 + r)
    g.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
for s in scs:
    eng.ScanGenomeCount(g, cp, s)
import time
fms = []
eng.ScanGenomeBegin(g, cp, scs[0])
t0 = time.perf_counter()
for i in range(passes):
    if i + 1 < passes:
        scs[(i + 1) & 1].chain_after(scs[i & 1])
        eng.ScanGenomeBegin(g, cp, scs[(i + 1) & 1])
    eng.ScanGenomeEndCount(g, cp, scs[i & 1])
    if i >= passes // 2:
        fms.append(scs[i & 1].stats().filter_ms)
dt = time.perf_counter() - t0
st = scs[0].stats()
print(f"{which}: step {dt/passes*1e3:.4f} ms  filter avg(second half) {sum(fms)/len(fms):.4f} ms  hits {st.hits} cand {st.candidates} kind {st.kernel_kind}")
