import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IPCR_JIT_ASYNC"] = "0"
from ipcr_amd import _lib, engine, primer, workloads
n_small = _lib.lib().ipcr_internal_small_launches
n_small.restype = ctypes.c_uint64
E = engine
pairs = workloads.c2_pairs()
cfg = E.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
eng = E.New(cfg)
cp = eng.CompilePanel(pairs)
sc = eng.NewSimulationScratch(cp)
rng = random.Random(1)
seq = bytes(rng.choice(b"ACGT") for _ in range(300000))
for segs, blocks in (("4", "512"), ("4", "0"), ("1", "512"), ("4", "512")):
    os.environ["IPCR_JIT_SEGMENTS"] = segs
    os.environ["IPCR_JIT_SEG_BLOCKS"] = blocks
    a = n_small()
    eng.SimulateCompiledWithScratch("s", seq, cp, sc)
    print(segs, blocks, "small launches:", n_small() - a, flush=True)
    g = E.Genome(3_000_000, 2)
    g.add_record("r", seq)
    a = n_small()
    eng.ScanGenome(g, cp, sc)
    print(segs, blocks, "resident: small launches:", n_small() - a, flush=True)
    g.close()
