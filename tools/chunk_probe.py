"""PCIe-inclusive rate of the drop-in entry point: ipcr_scan_chunk on host ASCII (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipcr_amd import engine, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
engine.lcg_fill_device(buf.data_ptr(), n, 0x5eed1234)   # the benchmark genome, generated on the device
seq = buf.cpu().numpy().tobytes()
del buf
eng = engine.New(engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12))
cp = eng.CompilePanel(workloads.c2_pairs())
sc = eng.NewSimulationScratch(cp)
for i in range(5):
    t0 = time.perf_counter(); eng.SimulateCompiledWithScratch("chr", seq, cp, sc); dt = time.perf_counter() - t0
    st = sc.stats()
    print(f"scan_chunk {n/1e6:.0f} Mb host ASCII: {dt*1e3:.2f} ms -> {n/dt/1e9:.2f} Gbases/s (pack {st.pack_ms:.3f} filter {st.filter_ms:.3f} verify {st.verify_ms:.3f} ms)", flush=True)
