"""C4 probe: multiplex panel (n TSV rows through ipcr-multiplex's unique self-pair rule) on one
resident 3 Gb genome; prints compile and scan times (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipcr_amd import engine, workloads, primer

E = engine
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nrec, reclen = int(sys.argv[2]) if len(sys.argv) > 2 else 24, int(sys.argv[3]) if len(sys.argv) > 3 else 125_000_000
g = E.Genome(nrec * reclen, nrec)
buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
pairs = workloads.c4_pairs(npairs)
for r in range(nrec):
    E.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed1234, r * reclen)
    for t in range(40):  # plant amplicons of 40 panel pairs per record
        p = pairs[(r * 40 + t) % npairs]
        start = 5000 + t * 100000
        buf[start:start + 20] = torch.tensor(list(p.Forward.encode()), dtype=torch.uint8)
        buf[start + 160:start + 180] = torch.tensor(list(primer.RevComp(p.Reverse)), dtype=torch.uint8)
    torch.cuda.synchronize()
    g.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
del buf
nlong = int(os.environ.get("C4_LONG", "0"))   # rows with 36-nt primers: patterns the seed index cannot key
if nlong:
    pairs = list(pairs) + [primer.Pair("long%d" % i, workloads.bench_primer(5000 + 2 * i, 36), workloads.bench_primer(5001 + 2 * i, 36), 128, 212) for i in range(nlong)]
cfg = E.Config(MaxMM=int(os.environ.get("C4_K", "2")), TerminalWindow=int(os.environ.get("C4_TW", "3")), MaxLen=2000, HitCap=10000, SeedLen=12)
eng = E.New(cfg)
t0 = time.time(); cp = eng.CompilePanel(pairs); print(f"CompilePanel {len(pairs)} pairs, {cp.num_patterns} patterns: {time.time()-t0:.2f} s", flush=True)
sc = eng.NewSimulationScratch(cp)
t0 = time.time(); n = eng.ScanGenomeCount(g, cp, sc); print(f"first scan (incl. kernel build): {time.time()-t0:.2f} s, products {n}", flush=True)
for i in range(3):
    t0 = time.time(); n = eng.ScanGenomeCount(g, cp, sc); dt = time.time() - t0
    st = sc.stats()
    print(f"scan: {dt*1e3:.1f} ms  leftover {st.leftover_patterns} patterns in {st.leftover_kernels} kernels  filter {st.filter_ms:.1f} ms verify {st.verify_ms:.2f} ms sort {st.sort_ms:.2f} join {st.join_ms:.2f} wait {st.wait_ms:.2f}  products {n} hits {st.hits} cand {st.candidates} kind {st.kernel_kind} -> {g.total_bases/dt/1e9:.1f} Gbases/s", flush=True)
