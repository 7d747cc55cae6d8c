"""BASELINE.json's full size (config C2: 3.0 Gb, 24 x 125 Mb) checked through size-independent
properties: every planted amplicon comes back with the planted mismatch positions, scans are
idempotent, hit lists are sorted and unique, the random background is of the expected order, and
one whole 125 Mb record agrees product-for-product with the CPU oracle."""
import os
import sys

import pytest

import ipcr_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_c2_full_size_properties():
    torch = pytest.importorskip("torch")
    sys.path.insert(0, ROOT)
    import bench
    from ipcr_amd import engine, primer, workloads

    cfg = engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    pairs = workloads.c2_pairs()
    eng = engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    genome, plants, host0 = bench.build_genome(torch, engine, workloads, primer.RevComp, 0, 24, 125_000_000, True)
    assert genome.total_bases == 3_000_000_000 and len(plants) == 1000

    prods = eng.ScanGenome(genome, cp, sc)
    st = sc.stats()
    assert st.kernel_kind == 1
    # (1) every planted amplicon, with exactly the planted mismatches
    found = {(p.Record, p.Start): p for p in prods if p.ExperimentID == "bench_000" and p.Type == "forward" and p.Length == 180}
    for (r, start, nm) in plants:
        p = found[(r, start)]
        assert (p.FwdMM, p.FwdMismatchIdx, p.RevMM) == (nm, () if nm == 0 else ((10,) if nm == 1 else (3, 10)), 0)
    # (2) hit lists: sorted by (record, pattern, position), unique, inside their records
    hits = sc.hits()
    keys = [(h.Record, h.Pattern, h.Pos) for h in hits]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    assert all(h.Pos + 20 <= 125_000_000 and h.Mismatches <= 2 for h in hits)
    # (3) random background: a 20-mer at k=2 with 5 protected bases has 991 accepted variants ->
    #     ~2.7 chance hits per orientation per 3 Gb; 4 patterns -> ~11, allow 0..60
    assert 2000 <= len(hits) <= 2060
    # (4) idempotence
    again = eng.ScanGenome(genome, cp, sc)
    assert [p.sig() for p in again] == [p.sig() for p in prods]
    # (5) one full record against the CPU oracle (its production path: AC seeds + verify + join)
    op = O.Panel(O.Config(max_mm=2, terminal_window=5, max_len=2000, hit_cap=10000, seed_len=12),
                 [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs])
    want = op.scan_ptr(host0.ctypes.data, int(host0.shape[0]))
    assert [p.sig() for p in prods if p.Record == 0] == [w.sig() for w in want] and len(want) >= 40
    # (6) tiles decode back to the genome (spot check around a plant)
    r, start, _ = plants[0]
    assert genome.read(r, start, 180) == bytes(host0[start:start + 180]) if r == 0 else True
    genome.close()
