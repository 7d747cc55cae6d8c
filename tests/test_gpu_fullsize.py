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


def test_config_c2_reference_n_full_size():
    """SURVEY 8(d)'s +N variant of C2 at its full size: 0.1 % of the positions of every record are 'N', in runs of 1-1000
    (bench.n_runs) -- for the reference that is forceFallback on every record (core/engine/compiled.go:185-190,238-258:
    FindMatches for all four orientations, the rc ones capped before their 5' window filter), for the device the kernel
    variant that scans the rc patterns unprotected and leaves the window to the host.  It is what every real genome and
    every ipcr_scan_chunk call run.  Same plants as the N-free genome; whole record 0 against the oracle with the
    default hit cap and with --hit-cap 0 (seeded path + halo rescue, collector order "automaton hits, then halo hits",
    compiled.go:211-232)."""
    torch = pytest.importorskip("torch")
    sys.path.insert(0, ROOT)
    import bench
    from ipcr_amd import engine, primer, workloads

    pairs = workloads.c2_pairs()
    genome, plants, host0 = bench.build_genome(torch, engine, workloads, primer.RevComp, 0, 24, 125_000_000, True, with_n=True)
    assert genome.total_bases == 3_000_000_000 and len(plants) == 1000
    n_count = int((host0 == 78).sum())
    assert 120_000 <= n_count <= 130_000                       # 0.1 % of 125 Mb
    assert all(genome.record_flags(r) & 1 for r in range(24))  # every record holds reset bytes
    opairs = [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs]
    for cap in (10000, 0):
        cfg = engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=cap, SeedLen=12)
        eng = engine.New(cfg)
        cp = eng.CompilePanel(pairs)
        sc = eng.NewSimulationScratch(cp)
        prods = eng.ScanGenome(genome, cp, sc)
        assert sc.stats().kernel_kind == 1
        if cap:   # the rc orientations are scanned without their protected window: patterns of their own
            assert cp.scanned_patterns(1) != cp.scanned_patterns(0)
        found = {(p.Record, p.Start): p for p in prods if p.ExperimentID == "bench_000" and p.Type == "forward" and p.Length == 180}
        for (r, start, nm) in plants:
            p = found[(r, start)]
            assert (p.FwdMM, p.FwdMismatchIdx, p.RevMM) == (nm, () if nm == 0 else ((10,) if nm == 1 else (3, 10)), 0)
        hits = sc.hits()
        keys = [(h.Record, h.Pattern, h.Pos) for h in hits]
        assert keys == sorted(keys) and len(set(keys)) == len(keys)
        assert all(h.Pos + 20 <= 125_000_000 and h.Mismatches <= 2 for h in hits)
        # unprotected rc patterns accept 1 + 60 + 1710 variants instead of 991: still a handful of chance hits per 3 Gb
        assert 2000 <= len(hits) <= 2100
        assert [p.sig() for p in eng.ScanGenome(genome, cp, sc)] == [p.sig() for p in prods]
        op = O.Panel(O.Config(max_mm=2, terminal_window=5, max_len=2000, hit_cap=cap, seed_len=12), opairs)
        want = op.scan_ptr(host0.ctypes.data, int(host0.shape[0]))
        assert [p.sig() for p in prods if p.Record == 0] == [w.sig() for w in want] and len(want) >= 40
        # the same genome under --chunk-size 4 Mb (ipcr_scan_genome_chunked: one sweep, 24 x 32 rolling windows, each joined as
        # its own call with its own reset flag -- asked on the device: with N in runs of up to 1000 most windows hold one, not
        # all): after the collector's rebasing and de-duplication the products are the whole-record ones (no cap bites here)
        import ctypes as C
        from ipcr_amd import _lib, cli
        chunked = eng.ScanGenomeChunked(genome, cp, sc, 4_000_000, 2020)
        w, nw = C.POINTER(_lib.ChunkWindow)(), C.c_int64()
        _lib.check(_lib.lib().ipcr_scratch_chunk_windows(sc._h, C.byref(w), C.byref(nw)))
        assert nw.value == 24 * 32 and all(w[i].end - w[i].start <= 4_000_000 for i in range(nw.value))
        dirty = sum(w[i].reset for i in range(nw.value))
        assert 0.5 * nw.value <= dirty <= nw.value
        coll = cli.Collector(1 << 20)
        kept = [q for q in (coll.add("g", p) for p in chunked) if q is not None]
        key = lambda p: (p.SequenceID, p.Start, p.End, p.ExperimentID, p.Type, p.FwdMM, p.RevMM, tuple(p.FwdMismatchIdx), tuple(p.RevMismatchIdx))
        assert sorted(map(key, kept)) == sorted(map(key, prods)) and len(chunked) >= len(kept)
        op.close()
        sc.close()
        cp.close()
    genome.close()


def test_config_c3_full_size_properties():
    """BASELINE.json configs[2]: 27F/1492R with IUPAC codes, k=3, --circular, 3.0 Gb.  Planted sites use
    concrete bases for the ambiguity codes (M -> A/C, Y -> C/T) and up to three substitutions outside the 3'
    window; one amplicon per record spans the origin.  Checked: every plant comes back with its mismatch
    positions, hit lists are sorted and unique, scans are idempotent, and a 20 Mb window of record 0 around two
    plants agrees with the CPU oracle (linear scan of the same bytes)."""
    torch = pytest.importorskip("torch")
    import random
    from ipcr_amd import engine, primer, workloads

    rng = random.Random(303)
    pairs = workloads.c3_pairs()
    fwd, rev = pairs[0].Forward, pairs[0].Reverse
    rc_rev = primer.RevComp(rev).decode()
    nrec, reclen = 24, 125_000_000
    cfg = engine.Config(MaxMM=3, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=True)
    eng = engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)

    def concrete(s):
        return "".join(rng.choice([b for b in "ACGT" if O.base_match(b, ch)]) for ch in s)

    genome = engine.Genome(nrec * reclen, nrec)
    buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
    plants, host0 = [], None
    for r in range(nrec):
        engine.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed3333 + r)
        for t in range(20):
            start = 1_000_000 + t * 6_000_000 + rng.randrange(1000)
            site = list(concrete(fwd))
            idx = sorted(rng.sample(range(0, len(fwd) - 3), rng.choice([0, 1, 2, 3])))   # 3' window stays clean
            for j in idx:
                site[j] = O.different_base(site[j]) if fwd[j] in "ACGT" else site[j]
            idx = tuple(j for j in idx if fwd[j] in "ACGT")
            buf[start:start + len(fwd)] = torch.tensor(list("".join(site).encode()), dtype=torch.uint8)
            buf[start + 400 - len(rc_rev):start + 400] = torch.tensor(list(concrete(rc_rev).encode()), dtype=torch.uint8)
            plants.append((r, start, idx))
        # origin-spanning amplicon: forward site near the end, reverse site near the start of the record
        buf[reclen - 150:reclen - 150 + len(fwd)] = torch.tensor(list(concrete(fwd).encode()), dtype=torch.uint8)
        buf[100:100 + len(rc_rev)] = torch.tensor(list(concrete(rc_rev).encode()), dtype=torch.uint8)
        torch.cuda.synchronize()
        if r == 0:
            host0 = buf[:20_000_000].cpu().numpy().copy()
        genome.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
    del buf

    prods = eng.ScanGenome(genome, cp, sc)
    assert sc.stats().kernel_kind == 1
    found = {(p.Record, p.Start): p for p in prods if p.ExperimentID == "16S" and p.Type == "forward" and p.Length == 400}
    for (r, start, idx) in plants:
        p = found[(r, start)]
        assert (p.FwdMM, p.FwdMismatchIdx, p.RevMM) == (len(idx), idx, 0), (r, start, idx, p)
    wraps = [p for p in prods if p.ExperimentID == "16S" and p.Type == "forward" and p.Start > p.End]
    assert {p.Record for p in wraps} == set(range(nrec))
    assert all(p.Start == reclen - 150 and p.End == 100 + len(rc_rev) and p.Length == 150 + 100 + len(rc_rev) for p in wraps)
    hits = sc.hits()
    keys = [(h.Record, h.Pattern, h.Pos) for h in hits]
    assert keys == sorted(keys) and len(set(keys)) == len(keys) and all(h.Mismatches <= 3 for h in hits)
    again = eng.ScanGenome(genome, cp, sc)
    assert [p.sig() for p in again] == [p.sig() for p in prods]
    # a 20 Mb prefix of record 0 against the CPU oracle (linear: the prefix is not a circular record)
    lin = engine.Config(MaxMM=3, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=False)
    want = O.simulate_batch(O.Config(max_mm=3, terminal_window=3, max_len=2000, hit_cap=10000, seed_len=12),
                            bytes(host0), [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs])
    got = engine.New(lin).SimulateBatch("chr1", bytes(host0), pairs)
    assert [g.sig() for g in got] == [w.sig() for w in want] and len(want) >= 4
    inside = [p.sig() for p in prods if p.Record == 0 and p.Start <= p.End and p.End <= 20_000_000 - 2000]
    assert inside == [w.sig() for w in want if w.end <= 20_000_000 - 2000]
    genome.close()


def test_config_c4_full_size_properties():
    """BASELINE.json configs[3] on one GPU's share: the 1024-row panel (3072 pairs / 4096 distinct patterns through
    ipcr-multiplex's self-pair rule) over a 3.0 Gb genome with 960 planted amplicons of 960 different pairs.
    Checked: seed-index filter in use, every plant found exactly (k=2 sites included), hits sorted and unique,
    idempotence, and the pattern-sharded run (pairs split in two halves, as a primer x genome tiling would) gives the
    same products."""
    torch = pytest.importorskip("torch")
    from ipcr_amd import engine, primer, workloads

    npairs, nrec, reclen = 1024, 24, 125_000_000
    pairs = workloads.c4_pairs(npairs)
    base = {p.ID: p for p in pairs[:npairs]}
    genome = engine.Genome(nrec * reclen, nrec)
    buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
    plants = {}
    for r in range(nrec):
        engine.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed1234, r * reclen)
        for t in range(40):
            p = pairs[(r * 40 + t) % npairs]
            start = 5000 + t * 100000
            site = list(p.Forward)
            nm = (r + t) % 3
            if nm >= 1:
                site[4] = workloads.different_base(site[4])
            if nm >= 2:
                site[11] = workloads.different_base(site[11])
            buf[start:start + 20] = torch.tensor(list("".join(site).encode()), dtype=torch.uint8)
            buf[start + 160:start + 180] = torch.tensor(list(primer.RevComp(p.Reverse)), dtype=torch.uint8)
            plants[(r, start, p.ID)] = nm
        torch.cuda.synchronize()
        genome.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
    del buf
    cfg = engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    assert cp.num_patterns == 4096
    sc = eng.NewSimulationScratch(cp)
    prods = eng.ScanGenome(genome, cp, sc)
    assert sc.stats().kernel_kind == 3
    found = {(p.Record, p.Start, p.ExperimentID): p for p in prods if p.Type == "forward" and p.Length == 180 and p.ExperimentID in base}
    for key, nm in plants.items():
        p = found[key]
        assert (p.FwdMM, p.FwdMismatchIdx, p.RevMM) == (nm, () if nm == 0 else ((4,) if nm == 1 else (4, 11)), 0)
    hits = sc.hits()
    keys = [(h.Record, h.Pattern, h.Pos) for h in hits]
    assert keys == sorted(keys) and len(set(keys)) == len(keys) and all(h.Mismatches <= 2 for h in hits)
    want = [p.sig() for p in prods]
    assert [p.sig() for p in eng.ScanGenome(genome, cp, sc)] == want
    # primer x genome tiling: two sub-panels over the same genome, products merged per pair
    half = []
    for sub in (pairs[0::2], pairs[1::2]):
        cps = eng.CompilePanel(sub)
        scs = eng.NewSimulationScratch(cps)
        half += [(p.Record, p.ExperimentID) + p.sig()[1:] for p in eng.ScanGenome(genome, cps, scs)]
        scs.close(); cps.close()
    assert sorted(half) == sorted((p.Record, p.ExperimentID) + p.sig()[1:] for p in prods)
    genome.close()


def test_config_c5_full_size_probe_rescan():
    """BASELINE.json configs[4]: pair + internal probe, k=2, 3.0 Gb.  1000 planted amplicons carry the probe at a
    known offset: on the + strand, as its reverse complement, with one or two substitutions, or not at all; the
    batched rescan (amplicon gather + probe kernel over all products of the scan) must report exactly that for
    every product, and agree with oligo.BestHit restated in the oracle on the amplicons of record 0."""
    torch = pytest.importorskip("torch")
    import random
    from ipcr_amd import _lib, engine, primer, workloads

    rng = random.Random(55)
    pairs = workloads.c2_pairs()
    fwd, rc_rev = pairs[0].Forward, primer.RevComp(pairs[0].Reverse).decode()
    probe = "TGGACCTTAGCAGGTCATTCAG"
    rc_probe = primer.RevComp(probe).decode()
    nrec, reclen = 24, 125_000_000
    genome = engine.Genome(nrec * reclen, nrec)
    buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
    plants, host0 = {}, None
    for r in range(nrec):
        engine.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed5555 + r)
        for t in range(42):
            if len(plants) >= 1000:
                break
            start = 2_000_000 + t * 2_900_000 + rng.randrange(500)
            kind = rng.choice(["plus", "minus", "mm1", "mm2", "none"])
            off = 30 + rng.randrange(90)
            buf[start:start + 20] = torch.tensor(list(fwd.encode()), dtype=torch.uint8)
            buf[start + 180 - 20:start + 180] = torch.tensor(list(rc_rev.encode()), dtype=torch.uint8)   # the pair allows 128..212
            site = list(probe if kind != "minus" else rc_probe)
            if kind in ("mm1", "mm2"):
                site[5] = workloads.different_base(site[5])
            if kind == "mm2":
                site[15] = workloads.different_base(site[15])
            if kind != "none":
                buf[start + off:start + off + len(probe)] = torch.tensor(list("".join(site).encode()), dtype=torch.uint8)
            plants[(r, start)] = (kind, off)
        torch.cuda.synchronize()
        if r == 0:
            host0 = buf.cpu().numpy().tobytes()
        genome.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
    del buf
    eng = engine.New(engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12))
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    prods = eng.ScanGenome(genome, cp, sc)
    mine = [i for i, p in enumerate(prods) if (p.Record, p.Start) in plants and p.Length == 180 and p.ExperimentID == "bench_000" and p.Type == "forward"]
    assert len(mine) == len(plants) == 1000
    for k in (0, 2):
        out = (_lib.ProbeHit * len(prods))()
        _lib.check(_lib.lib().ipcr_probe_products(sc._h, genome._h, probe.encode(), k, out, len(prods)))
        for i in mine:
            kind, off = plants[(prods[i].Record, prods[i].Start)]
            h = out[i]
            expect_found = kind in ("plus", "minus") or (k == 2 and kind in ("mm1", "mm2"))
            assert bool(h.found) == expect_found, (kind, k, h.found)
            if expect_found:
                assert (chr(h.strand), h.pos, h.mm) == ("-" if kind == "minus" else "+", off, {"plus": 0, "minus": 0, "mm1": 1, "mm2": 2}[kind])
        for i, p in enumerate(prods):                 # every product of record 0 (plants and background) vs the oracle
            if p.Record != 0 or p.Start > p.End:
                continue
            w = O.best_hit(host0[p.Start:p.End], probe, k)
            assert (bool(out[i].found), chr(out[i].strand) if out[i].found else "", out[i].pos, out[i].mm) == \
                (w.found, w.strand, w.pos if w.found else 0, w.mm if w.found else 0)
    genome.close()


def test_resident_genome_beyond_2_to_32_bases():
    """More than 2^32 bases resident at once: 36 records x 125 Mb = 4.5 Gb (2.25 GB of tiles + rst) -- padded coordinates,
    tile word indices and the hit records' positions are 64-bit, grid sizes and counters must not wrap on the way.
    Amplicons planted in every record, the last ones beyond base 4.3e9; every plant comes back with its mismatch
    positions, hit lists are sorted and unique and lie inside their records, the scan is idempotent, and the LAST record
    agrees product for product with the CPU oracle."""
    torch = pytest.importorskip("torch")
    from ipcr_amd import engine, primer, workloads

    nrec, reclen = 36, 125_000_000
    assert nrec * reclen > (1 << 32)
    pairs = workloads.c2_pairs()
    fwd, rc_rev = pairs[0].Forward, primer.RevComp(pairs[0].Reverse).decode()
    genome = engine.Genome(nrec * reclen, nrec)
    buf = torch.empty(reclen, dtype=torch.uint8, device="cuda:0")
    plants, host_last = {}, None
    for r in range(nrec):
        engine.lcg_fill_device(buf.data_ptr(), reclen, 0x5eed7000 + r)       # (one LCG stream per record: its period is 2^32)
        for t in range(6):
            start = 3_000_000 + t * 24_000_000 + 17 * r
            nm = (r + t) % 3
            site = list(fwd)
            if nm >= 1:
                site[10] = workloads.different_base(site[10])
            if nm >= 2:
                site[3] = workloads.different_base(site[3])
            buf[start:start + 20] = torch.tensor(list("".join(site).encode()), dtype=torch.uint8)
            buf[start + 160:start + 180] = torch.tensor(list(rc_rev.encode()), dtype=torch.uint8)
            plants[(r, start)] = nm
        buf[reclen - 180:reclen - 160] = torch.tensor(list(fwd.encode()), dtype=torch.uint8)     # flush with the record's end
        buf[reclen - 20:reclen] = torch.tensor(list(rc_rev.encode()), dtype=torch.uint8)
        plants[(r, reclen - 180)] = 0
        torch.cuda.synchronize()
        if r == nrec - 1:
            host_last = buf.cpu().numpy().copy()
        genome.add_record_device("chr%d" % (r + 1), buf.data_ptr(), reclen)
    del buf
    assert genome.total_bases == nrec * reclen > (1 << 32)
    cfg = engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    prods = eng.ScanGenome(genome, cp, sc)
    st = sc.stats()
    assert st.kernel_kind == 1 and st.bases == nrec * reclen
    found = {(p.Record, p.Start): p for p in prods if p.ExperimentID == "bench_000" and p.Type == "forward" and p.Length == 180}
    for (r, start), nm in plants.items():
        p = found[(r, start)]
        assert (p.FwdMM, p.FwdMismatchIdx, p.RevMM) == (nm, () if nm == 0 else ((10,) if nm == 1 else (3, 10)), 0), (r, start)
    hits = sc.hits()
    keys = [(h.Record, h.Pattern, h.Pos) for h in hits]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    assert all(0 <= h.Record < nrec and h.Pos + 20 <= reclen and h.Mismatches <= 2 for h in hits)
    assert {h.Record for h in hits} == set(range(nrec))
    assert [p.sig() for p in eng.ScanGenome(genome, cp, sc)] == [p.sig() for p in prods]
    op = O.Panel(O.Config(max_mm=2, terminal_window=5, max_len=2000, hit_cap=10000, seed_len=12),
                 [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs])
    want = op.scan_ptr(host_last.ctypes.data, int(host_last.shape[0]))
    assert [p.sig() for p in prods if p.Record == nrec - 1] == [w.sig() for w in want] and len(want) >= 7
    assert genome.read(nrec - 1, reclen - 180, 180) == bytes(host_last[reclen - 180:])        # tiles of the last record decode back
    # the seed-index kernel walks the same coordinates (column pairs beyond 2^19, positions beyond 2^32)
    rows = workloads.c4_pairs(128)
    cp4 = engine.New(engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)).CompilePanel(rows)
    sc4 = eng.NewSimulationScratch(cp4)
    prods4 = engine.New(engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)).ScanGenome(genome, cp4, sc4)
    assert sc4.stats().kernel_kind == 3
    found4 = {(p.Record, p.Start) for p in prods4 if p.ExperimentID == "bench_000" and p.Type == "forward" and p.Length == 180}
    assert all(key in found4 for key, nm in plants.items())          # (window 3: every planted site passes here too)
    op.close(); sc4.close(); cp4.close(); sc.close(); cp.close()
    genome.close()
