#!/usr/bin/env python3
"""Load + scan time of a fragmented assembly (many short records): tools/many_records_probe.py [records] [len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ipcr_amd import engine, workloads

nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reclen = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
buf = torch.empty(nrec * reclen, dtype=torch.uint8, device="cuda:0")
engine.lcg_fill_device(buf.data_ptr(), nrec * reclen, 0x5eed9999)
seq = buf.cpu().numpy()
del buf
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "many_%d.fa" % os.getpid())
with open(path, "wb") as fh:
    for r in range(nrec):
        fh.write(b">ctg%d\n" % r)
        rec = seq[r * reclen:(r + 1) * reclen]
        full = (reclen // 80) * 80
        body = np.empty((full // 80, 81), dtype=np.uint8)
        body[:, :80] = rec[:full].reshape(-1, 80)
        body[:, 80] = 10
        fh.write(body.tobytes())
        if full < reclen:
            fh.write(rec[full:].tobytes() + b"\n")
eng = engine.New(engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12))
cp = eng.CompilePanel(workloads.c2_pairs())
sc = eng.NewSimulationScratch(cp)
for attempt in range(2):
    t0 = time.perf_counter()
    g = engine.Genome(nrec * (reclen + 8192 * 2) + (1 << 20), max_records=nrec + 8)
    n = g.add_fasta(path)
    t1 = time.perf_counter()
    k = eng.ScanGenomeCount(g, cp, sc)
    t2 = time.perf_counter()
    st = sc.stats()
    print(f"{n} records x {reclen}: load {1e3*(t1-t0):.1f} ms, scan {1e3*(t2-t1):.2f} ms (filter {st.filter_ms:.3f}, sort {st.sort_ms:.3f}, join {st.join_ms:.3f}), "
          f"tiles {g.tile_bytes/1e6:.0f} MB, products {k}", flush=True)
    g.close()
os.unlink(path)
