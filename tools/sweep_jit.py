"""Sweep the specialised-filter generator knobs on one resident 3 Gb genome (dev tool)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipcr_amd import engine, workloads

E = engine
nrec, reclen = 24, 125_000_000
g = E.Genome(nrec * reclen, nrec)
buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
for r in range(nrec):
    E.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed1234, r * reclen)
    g.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
del buf
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    cfg, pairs = E.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12), workloads.c2_pairs()
else:
    cfg, pairs = E.Config(MaxMM=3, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=True), workloads.c3_pairs()
eng = E.New(cfg)
for kv in os.environ.get("SWEEP_ENV", "").split():   # extra generator knobs for the whole sweep: "IPCR_JIT_FILTER_LEN=16 ..."
    k_, v_ = kv.split("=", 1)
    os.environ[k_] = v_
configs = [tuple(int(x) for x in c.split(',')) for c in os.environ['SWEEP_CONFIGS'].split(';')] if os.environ.get('SWEEP_CONFIGS') else [(2, 2, 4), (3, 2, 4), (1, 2, 4), (2, 2, 2), (3, 2, 2), (2, 3, 2), (3, 3, 2), (2, 3, 1), (4, 1, 4), (3, 1, 4), (2, 2, 1), (3, 2, 1)]
for (d, w, wg) in configs:
    os.environ["IPCR_JIT_DEPTH"], os.environ["IPCR_JIT_WAVES"], os.environ["IPCR_JIT_WG"] = str(d), str(w), str(wg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    ts = []
    warm = int(os.environ.get("SWEEP_WARM", "150"))   # the first ~50 sweeps after idle run 15-25 % slower (clock ramp)
    for i in range(warm + 40):
        eng.ScanGenomeCount(g, cp, sc)
        if i >= warm:
            ts.append(sc.stats().filter_ms)
    st = sc.stats()
    med = statistics.median(ts)
    print(f"{which} depth={d} waves={w} wg={wg}: filter_ms median {med:.4f} min {min(ts):.4f}  -> {g.total_bases/med/1e6:.0f} Gb/s, {0.375*g.total_bases/med/1e6:.0f} GB/s  kind={st.kernel_kind} cand={st.candidates} hits={st.hits}", flush=True)
    sc.close(); cp.close()
