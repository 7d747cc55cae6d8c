import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipcr_amd import engine, primer, workloads, _lib

def t(msg, f):
    t0 = time.time(); r = f(); print(f"{msg}: {(time.time()-t0)*1e3:.1f} ms", flush=True); return r

E = engine
cfg = E.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
pairs = workloads.c2_pairs()
eng = E.New(cfg)
cp = t("CompilePanel", lambda: eng.CompilePanel(pairs))
sc = t("scratch", lambda: eng.NewSimulationScratch(cp))
seq = b"ACGT" * 1000
for i in range(3):
    t("scan_chunk small (spec)", lambda: eng.SimulateCompiledWithScratch("s", seq, cp, sc))
cp2 = eng.CompilePanel(pairs); cp2.set_specialize(False)
sc2 = t("scratch2", lambda: eng.NewSimulationScratch(cp2))
for i in range(3):
    t("scan_chunk small (generic)", lambda: eng.SimulateCompiledWithScratch("s", seq, cp2, sc2))
    st = sc2.stats(); print("   stats filter_ms %.3f verify_ms %.3f pack %.3f total %.3f" % (st.filter_ms, st.verify_ms, st.pack_ms, st.total_ms))

# big genome
nrec, reclen = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000
g = t("genome_create", lambda: E.Genome(nrec * reclen, nrec))
buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
for r in range(nrec):
    E.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed1234 + r)
    g.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
print("pack_ms total", g.pack_ms, "bases", g.total_bases, "tile_bytes", g.tile_bytes, flush=True)
for name, c, s in (("spec", cp, sc), ("generic", cp2, sc2)):
    for i in range(3 if name == "spec" else 1):
        n = t(f"ScanGenome {name}", lambda: eng.ScanGenomeCount(g, c, s))
        st = s.stats()
        print(f"   products {n} hits {st.hits} cand {st.candidates} kind {st.kernel_kind} filter_ms {st.filter_ms:.3f} verify_ms {st.verify_ms:.3f} total_ms {st.total_ms:.3f}", flush=True)
        print(f"   Gbases/s (filter) {st.bases/st.filter_ms/1e6:.1f}  tile GB/s {st.tile_bytes/st.filter_ms/1e6:.1f}", flush=True)
