"""GPU parity: the HIP path (through the C ABI, via the ipcr_amd mirror of the reference API)
against the CPU oracle on the same inputs.  Bit-exact, including emission order."""
import random
from collections import Counter

import pytest

import ipcr_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ipcr_amd import _lib, engine, oligo, primer, probe
    assert _lib.lib().ipcr_device_count() > 0, "needs a HIP device"

    class NS:
        pass
    ns = NS()
    ns.lib, ns.engine, ns.primer, ns.oligo, ns.probe = _lib, engine, primer, oligo, probe
    return ns


def ocfg(c):
    return O.Config(max_mm=c.MaxMM, terminal_window=c.TerminalWindow, min_len=c.MinLen, max_len=c.MaxLen,
                    hit_cap=c.HitCap, seed_len=c.SeedLen, circular=c.Circular)


def opairs(pairs):
    return [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs]


def check(hip, cfg, seq, pairs, specialize=None):
    eng = hip.engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    if specialize is not None:
        cp.set_specialize(specialize)
    got = eng.SimulateCompiled("seq", seq, cp)
    want = O.simulate_batch(ocfg(cfg), seq, opairs(pairs))
    assert [g.sig() for g in got] == [w.sig() for w in want]
    op = O.Panel(ocfg(cfg), opairs(pairs))
    for i in range(len(pairs)):
        for w in "ABab":
            assert cp.have(i, w) == op.have(i, w)
    cp.close()
    return got


# ---- the reference's own engine tests, run against the HIP engine ----------------------------

def test_simulate_minimal(hip):  # core/engine/engine_test.go:11-35
    P = hip.primer.Pair
    got = hip.engine.New(hip.engine.Config()).Simulate("dummySeq", b"ACGTACGTACGT", P("test", "ACG", "ACG"))
    assert got and (got[0].Start, got[0].End, got[0].Length) == (0, 12, 12)
    assert len(got) == 12


def test_length_filtering_and_type(hip):  # engine_test.go:38-78
    P = hip.primer.Pair
    eng = hip.engine.New(hip.engine.Config())
    hits = eng.Simulate("seq", b"ACGTACGTACGT", P("t", "ACG", "ACG", 10, 12))
    assert hits and all(10 <= p.Length <= 12 for p in hits)
    assert eng.Simulate("seq", b"ACGTACGTACGT", P("t2", "ACG", "ACG", 5, 7)) == []


def test_revcomp_product(hip):  # engine_test.go:81-101
    hits = hip.engine.New(hip.engine.Config()).Simulate("s", b"TTTACGACGTAAA", hip.primer.Pair("rev", "ACG", "TTT"))
    assert any(h.Type == "revcomp" for h in hits)


def test_circular_amplicon(hip):  # engine_test.go:104-129
    E, P = hip.engine, hip.primer.Pair
    assert E.New(E.Config(Circular=False)).Simulate("seq1", b"TGACAAG", P("p1", "AG", "TC")) == []
    hits = E.New(E.Config(Circular=True)).Simulate("seq1", b"TGACAAG", P("p1", "AG", "TC"))
    assert len(hits) == 1 and hits[0].Start > hits[0].End
    assert hits[0].Length == 7 - hits[0].Start + hits[0].End


def test_seeded_mismatch_cases(hip):  # engine_test.go:131-168
    E, P = hip.engine, hip.primer.Pair
    hits = E.New(E.Config(MaxMM=1, TerminalWindow=3, SeedLen=12, MinLen=10)).Simulate(
        "seq", b"CAGTACAAAAAAGGTACC", P("seed-mm", "AAGTAC", "GGTACC"))
    assert len(hits) == 1 and hits[0].FwdMM == 1 and hits[0].FwdMismatchIdx == (0,)
    eng = E.New(E.Config(MaxMM=1, TerminalWindow=0, SeedLen=12, MinLen=10))
    cp = eng.CompilePanel([P("seed-mm-no-tw", "AAGTAC", "GGTACC")])
    assert cp.have(0, "A")
    assert len(eng.Simulate("seq", b"CAGTACAAAAAAGGTACC", P("seed-mm-no-tw", "AAGTAC", "GGTACC"))) == 1


ORACLE_CASES = [  # core/engine/approx_seed_oracle_test.go:95-176
    ("TTTACGTACAAAAGGTACCTTT", [("forward_exact", "ACGTAC", "GGTACC")]),
    ("TTTGGTACCAAAAGTACGTTTT", [("revcomp_exact", "ACGTAC", "GGTACC")]),
    ("TTTTCGTACAAAAGGTACCTTT", [("forward_mismatch_5prime", "ACGTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTT", [("primer_ry", "ACRTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTT", [("primer_internal_n", "ACNTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTT", [("primer_3prime_n", "ACGTAN", "GGTACC")]),
    ("TTTNCGTACAAAAGGTACCTTT", [("reference_n", "ACGTAC", "GGTACC")]),
    ("TTTacgtacAAAAGGTACCTTT", [("lowercase_reference", "ACGTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTTGGGGGGGGGG", [("panel_hit", "ACGTAC", "GGTACC"), ("panel_decoy", "TTAACC", "CCAATT")]),
]


@pytest.mark.parametrize("spec", [False, True])
def test_oracle_matrix(hip, spec):  # approx_seed_oracle_test.go:88-203 (9 sequences x 9 configs)
    for seq, ps in ORACLE_CASES:
        pairs = [hip.primer.Pair(*p) for p in ps]
        for k in (0, 1, 2):
            for tw in (0, 1, 3):
                cfg = hip.engine.Config(MaxMM=k, TerminalWindow=tw, MinLen=1, MaxLen=100, SeedLen=12)
                check(hip, cfg, seq.encode(), pairs, specialize=spec)


def test_length_boundary_circular_self(hip):  # approx_seed_oracle_test.go:205-259
    E, P = hip.engine, hip.primer.Pair
    for mn, mx in [(16, 16), (17, 100), (1, 15)]:
        check(hip, E.Config(MinLen=mn, MaxLen=mx, SeedLen=12), b"TTTACGTACAAAAGGTACCTTT", [P("length", "ACGTAC", "GGTACC")])
    for circ in (False, True):
        check(hip, E.Config(Circular=circ), b"TGACAAG", [P("circular", "AG", "TC")])
    pairs = hip.primer.SelfPairs([hip.primer.Oligo("self", "ACGTAC")])
    for k, tw in [(0, 0), (1, 3), (2, 0)]:
        check(hip, E.Config(MaxMM=k, TerminalWindow=tw, MinLen=1, MaxLen=100, SeedLen=12), b"TTTACGTACAAAAGTACGTTTT", pairs)


def test_halo_case(hip):  # non_acgt_halo_test.go:32-52
    got = check(hip, hip.engine.Config(MaxMM=1, TerminalWindow=0, MinLen=1, MaxLen=100, SeedLen=6),
                b"TTTACNTACAAAAGGTACCTTT", [hip.primer.Pair("reference_n_inside_seed", "ACGTAC", "GGTACC")])
    assert len(got) >= 1


@pytest.mark.parametrize("spec", [False, True])
@pytest.mark.parametrize("mutate,refn,k,n", [(True, True, 2, 12), (True, False, 1, 16), (False, False, 0, 64), (False, True, 1, 16)])
def test_bench_fixtures(hip, spec, mutate, refn, k, n):  # performance_gate_test.go:51-59, performance_benchmark_test.go:155-213
    seq, op = O.make_bench_fixture(n, 250000 if n >= 16 else 20000, mutate, refn)
    pairs = [hip.primer.Pair(p.id, p.forward, p.reverse, p.min_product, p.max_product) for p in op]
    got = check(hip, hip.engine.Config(MaxMM=k, TerminalWindow=0, MinLen=100, MaxLen=240, SeedLen=12), seq, pairs, specialize=spec)
    assert len(got) >= n


# ---- randomised differential tests -------------------------------------------------------------

def rand_case(rng, n, with_junk):
    alpha = "ACGT"
    seq = [rng.choice(alpha) for _ in range(n)]
    if with_junk:
        for _ in range(rng.randint(0, 6)):
            p = rng.randrange(n)
            run = rng.randint(1, 12)
            ch = rng.choice("NNNRYacgtn")
            for i in range(p, min(n, p + run)):
                seq[i] = ch
    return seq


def plant(rng, seq, primer_seq, pos, nmut):
    s = list(primer_seq)
    concrete = []
    for ch in s:
        opts = [b for b in "ACGT" if O.base_match(b, ch)]
        concrete.append(rng.choice(opts))
    for _ in range(nmut):
        j = rng.randrange(len(concrete))
        concrete[j] = O.different_base(concrete[j])
    seq[pos:pos + len(concrete)] = concrete


@pytest.mark.parametrize("spec", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_random_differential(hip, spec, seed):
    rng = random.Random(1234 + seed)
    E, P = hip.engine, hip.primer.Pair
    for it in range(10 if not spec else 6):  # every specialised panel is a hiprtc compile (seconds)
        n = rng.choice([60, 300, 5000, 70000])
        seq = rand_case(rng, n, with_junk=rng.random() < 0.6)
        npairs = rng.randint(1, 4)
        pairs = []
        lmin = 4 if n <= 300 else (8 if n <= 5000 else 14)  # keeps the product lists small
        for i in range(npairs):
            def mk():
                L = rng.randint(lmin, 30)
                s = [rng.choice("ACGT") for _ in range(L)]
                for _ in range(rng.choice([0, 0, 1, 2])):
                    s[rng.randrange(L)] = rng.choice("RYSWKMBDHVN")
                return "".join(s)
            pairs.append(P("p%d" % i, mk(), mk(), rng.choice([0, 0, 20]), rng.choice([0, 0, 400])))
        # plant amplicons so products exist
        for p in pairs:
            for _ in range(rng.randint(1, 3)):
                a = rng.randrange(0, max(1, n - 200))
                ln = rng.randint(len(p.Forward) + len(p.Reverse), 150)
                if a + ln > n:
                    continue
                plant(rng, seq, p.Forward, a, rng.choice([0, 0, 1, 2]))
                rc = O.revcomp(p.Reverse).decode()
                plant(rng, seq, rc, a + ln - len(rc), rng.choice([0, 0, 1]))
        cfg = E.Config(MaxMM=rng.choice([0, 1, 2, 3] if n <= 5000 else [0, 1, 2]), TerminalWindow=rng.choice([0, 1, 3, 5]),
                       MinLen=rng.choice([0, 10]), MaxLen=rng.choice([0, 200, 2000]),
                       HitCap=rng.choice([0, 0, 3, 10000]), SeedLen=rng.choice([0, 12, 6, -1]),
                       Circular=rng.random() < 0.3)
        if rng.random() < 0.5:
            pairs = hip.primer.AddSelfPairs(pairs)
        check(hip, cfg, "".join(seq).encode(), pairs, specialize=spec)


@pytest.mark.parametrize("roll", [0, 1, 2])
def test_windows_across_strand_and_block_ends(hip, monkeypatch, roll):
    """sites planted so that their windows start in the last rows of a strand (p % 128 in 100..127: the wrap rows, read
    from the stashed head quads of the neighbour column) and across block ends (p % 8192 near 8191: lane 63's neighbour
    is column 0 of the next block), forward and reverse, 0-2 mismatches, with junk bytes; the specialised filter in
    its default form (iteration 0, main loop, static epilogue), as one rolled loop over all quads (IPCR_JIT_ROLL=1) and
    with a 24-slot window and one rare-branch test per quad (IPCR_JIT_MERGE=1; jit.cpp); vs the oracle"""
    monkeypatch.setenv("IPCR_JIT_ROLL", str(roll & 1))   # read when the kernel's source is generated (first scan)
    monkeypatch.setenv("IPCR_JIT_MERGE", str(roll >> 1))  # 2: a 24-slot register window, one rare-branch test per row quad
    rng = random.Random(4242 + roll)
    E, P = hip.engine, hip.primer.Pair
    for n, k, tw in ((34000, 2, 5), (18000, 3, 3), (34000, 0, 0)):
        seq = rand_case(rng, n, with_junk=True)
        pairs = [P("a", "ACGTTGCATGGATCCTAACG", "TTGACCGTAGGCATTCAGGA", 0, 0), P("b", "AGAGTTTGATCMTGGCTCAG", "TACGGYTACCTTGTTAYGACTT", 0, 0)]
        starts = [s0 + r for s0 in range(0, n - 9000, 8192) for r in (100, 109, 118, 127, 8064 + 120, 8191 - 10, 8191)]
        for i, a in enumerate(starts):
            p = pairs[i % 2]
            ln = rng.randint(60, 150)
            if a + ln > n:
                continue
            plant(rng, seq, p.Forward, a, rng.choice([0, 1, 2]) if k else 0)
            rc = O.revcomp(p.Reverse).decode()
            plant(rng, seq, rc, a + ln - len(rc), rng.choice([0, 0, 1]) if k else 0)
        cfg = E.Config(MaxMM=k, TerminalWindow=tw, MinLen=0, MaxLen=400, HitCap=0, SeedLen=12)
        check(hip, cfg, "".join(seq).encode(), hip.primer.AddSelfPairs(pairs), specialize=True)


def test_table_driven_filter_across_strand_column_and_block_ends(hip):
    """the table-driven filter (a wave per row quad: four start rows, 16-byte loads, the pattern's masks in scalar registers;
    kernels.hip: filter_generic_quad_kernel) on sites whose windows start in the last rows of a strand (the walk continues
    in the next strand: the same words one bit down), in the last strand of a column (the next column's bit 0 comes in on
    top) and of a block (lane 63's neighbour is column 0 of the next block: p near 262144), k = 0..3 with and without a
    terminal window (counter depths 1..4), IUPAC primers, junk bytes; vs the oracle.  Larger k: test_large_k_on_every_kernel"""
    rng = random.Random(777)
    E, P = hip.engine, hip.primer.Pair
    n = 270000
    for k, tw in ((2, 5), (3, 3), (0, 0), (1, 0)):
        seq = rand_case(rng, n, with_junk=True)
        pairs = [P("a", "ACGTTGCATGGATCCTAACG", "TTGACCGTAGGCATTCAGGA", 0, 0), P("b", "AGAGTTTGATCMTGGCTCAG", "TACGGYTACCTTGTTAYGACTT", 0, 0)]
        starts = [s0 + r for s0 in (0, 3 * 4096, 131072, 262144 - 4096) for r in (100, 109, 118, 125, 126, 127, 4096 - 128 + 120, 4095 - 10, 4095)]
        starts += [262144 - d for d in (1, 5, 10, 19, 20, 21, 40, 127, 128)]
        for i, a in enumerate(sorted(set(starts))):
            p = pairs[i % 2]
            ln = rng.randint(60, 150)
            if a + ln > n:
                continue
            plant(rng, seq, p.Forward, a, rng.choice([0, 1, 2]) if k else 0)
            rc = O.revcomp(p.Reverse).decode()
            plant(rng, seq, rc, a + ln - len(rc), rng.choice([0, 0, 1]) if k else 0)
        cfg = E.Config(MaxMM=k, TerminalWindow=tw, MinLen=0, MaxLen=400, HitCap=0, SeedLen=12)
        got = check(hip, cfg, "".join(seq).encode(), hip.primer.AddSelfPairs(pairs), specialize=False)
        assert len(got) >= 10


@pytest.mark.parametrize("seed", range(4))
def test_long_primers_take_the_specialised_filter(hip, seed):
    """primers of 33..128 nt: the specialised kernel filters on the 20 positions next to the protected end, every
    survivor goes through the global queue and the stand-alone verifier (the wave's own verifier takes <= 32 nt);
    mixed with short primers in one panel; vs the oracle, and the kernel kind is checked"""
    rng = random.Random(4100 + seed)
    E, P = hip.engine, hip.primer.Pair
    for it in range(3):
        n = rng.choice([3000, 40000])
        seq = rand_case(rng, n, with_junk=rng.random() < 0.5)
        pairs = []
        for i in range(rng.randint(1, 3)):
            def mk(lo, hi):
                L = rng.randint(lo, hi)
                s = [rng.choice("ACGT") for _ in range(L)]
                for _ in range(rng.choice([0, 1, 2])):
                    s[rng.randrange(L)] = rng.choice("RYSWKMBDHVN")
                return "".join(s)
            pairs.append(P("p%d" % i, mk(33, 128), mk(*rng.choice([(33, 128), (16, 30), (65, 128)]))))
        for p in pairs:
            for _ in range(2):
                a = rng.randrange(0, n - 700)
                ln = rng.randint(len(p.Forward) + len(p.Reverse), 600)
                plant(rng, seq, p.Forward, a, rng.choice([0, 1, 2, 3]))
                rc = O.revcomp(p.Reverse).decode()
                plant(rng, seq, rc, a + ln - len(rc), rng.choice([0, 1]))
        cfg = E.Config(MaxMM=rng.choice([0, 1, 2, 3]), TerminalWindow=rng.choice([0, 3, 5]), MaxLen=rng.choice([0, 1000]),
                       HitCap=rng.choice([0, 10000]), SeedLen=rng.choice([0, 12]), Circular=rng.random() < 0.3)
        if rng.random() < 0.5:
            pairs = hip.primer.AddSelfPairs(pairs)
        b = "".join(seq).encode()
        check(hip, cfg, b, pairs)
        eng = E.New(cfg)
        cp = eng.CompilePanel(pairs)
        sc = eng.NewSimulationScratch(cp)
        eng.SimulateCompiledWithScratch("s", b, cp, sc)
        assert sc.stats().kernel_kind == 1, "a panel with long primers fell back to the table-driven filter"
        sc.close()
        cp.close()


def test_hit_cap_quirks(hip):
    """HitCap truncation per orientation, incl. the reference's cap-before-5'-filter order on the
    FindMatches path (core/engine/compiled.go:249-256) with and without non-ACGT bytes."""
    E, P = hip.engine, hip.primer.Pair
    rng = random.Random(7)
    base = "ACGTTGCA" * 40
    for junk in ("", "N"):
        for cap in (1, 2, 5, 0):
            for k, tw in [(1, 2), (2, 3), (0, 3), (1, 0)]:
                seq = (base[:100] + junk + base[100:]).encode()
                cfg = E.Config(MaxMM=k, TerminalWindow=tw, MaxLen=60, HitCap=cap, SeedLen=rng.choice([4, 12, -1]))
                check(hip, cfg, seq, [P("q", "ACGTTG", "TGCAAC"), P("r", "CGTT", "GCAA")])


def test_host_packed_chunks_take_the_pattern_set_their_bytes_ask_for(hip, monkeypatch):
    """A chunk the HOST packs is scanned knowing whether it holds a reset byte: without one every orientation keeps its
    window on the device (what the reference's seeded path does, compiled.go:185-258), with one the rc orientations are
    scanned raw and the host applies the window after the cap (compiled.go:249-256).  A chunk the device packs takes the
    second form whatever it holds.  The cap quirks, the halo case and random panels under the host's packer, vs the oracle;
    and which form ran is read back from the statistics."""
    from ipcr_amd import workloads
    E = hip.engine
    monkeypatch.setenv("IPCR_CHUNK_HOSTPACK", "1")
    test_hit_cap_quirks(hip)
    test_halo_case(hip)
    test_random_differential(hip, False, 2)
    pairs = workloads.c2_pairs()
    cfg = E.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = E.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    rng = random.Random(31)
    seq = rand_case(rng, 300_000, with_junk=False)
    for a in (1000, 150_000, 299_000):
        plant(rng, seq, pairs[0].Forward, a, 1)
        plant(rng, seq, O.revcomp(pairs[0].Reverse).decode(), a + 160, 1)    # (the pair takes products of 128..212)
    clean = "".join(seq).encode()
    dirty = clean[:70_000] + b"N" + clean[70_001:]
    lower = clean[:70_000] + b"acgt" + clean[70_004:]        # lower case is not a reset byte (hostpack.cpp: flag bit 1)
    # IPCR_CHUNK_BAR: the packer writes the code planes into device memory through the PCIe BAR (default where the BAR is
    # large) / into pinned slabs that a copy operation takes over (IPCR_CHUNK_BAR=0, and every host without a large BAR);
    # IPCR_CHUNK_SKIP_INV=0: the invalid-bit plane crosses the link even for a slice of ACGT only
    import ctypes
    how = hip.lib.lib().ipcr_internal_device_bar
    how.restype, how.argtypes = ctypes.c_int32, [ctypes.c_int32]
    # 2: the whole of device memory is mapped (large BAR) and the runtime's HDP flush register was found; 0: no large BAR.  Never
    # 1 by default: without the flush register the library does not write through the BAR
    assert how(0) in (0, 2)
    for bar, skip in (("1", "1"), ("0", "1"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("IPCR_CHUNK_BAR", bar)
        monkeypatch.setenv("IPCR_CHUNK_SKIP_INV", skip)
        for name, b, env, want_set in (("clean", clean, "1", 0), ("dirty", dirty, "1", 1), ("lower", lower, "1", 0),
                                       ("device-packed", clean, "0", 1), ("clean", clean, "1", 0), ("dirty", dirty, "1", 1)):
            monkeypatch.setenv("IPCR_CHUNK_HOSTPACK", env)
            got = [p.sig() for p in eng.SimulateCompiledWithScratch("s", b, cp, sc)]
            want = [w.sig() for w in O.simulate_batch(ocfg(cfg), b, opairs(pairs))]
            assert got == want and len(want) >= 3, (name, bar, skip)
            assert sc.stats().pattern_set == want_set, (name, bar, skip)
        # lengths around the word, strand and column ends, the amplicon at the very end: the invalid bits behind the record's
        # end are made on the device when the plane does not cross the link
        for n in (160, 4095, 4096, 4097, 12_289, 131_071):
            tail = bytearray(clean[:n])
            tail[n - 180:n - 160] = pairs[0].Forward.encode()
            tail[n - 20:n] = O.revcomp(pairs[0].Reverse)
            tail = bytes(tail) if n >= 180 else clean[:n]
            got = [p.sig() for p in eng.SimulateCompiledWithScratch("s", tail, cp, sc)]
            want = [w.sig() for w in O.simulate_batch(ocfg(cfg), tail, opairs(pairs))]
            assert got == want and (n < 180 or len(want) >= 1), (n, bar, skip)
    # a lone worker's 20 Mb record: the pack pool fills three groups of columns (2048 columns = 8 388 608 bases each), the first
    # clean, the second with a run of N (its invalid-bit plane follows by DMA), the third with lower case (invalid + reset planes);
    # amplicons across both group boundaries
    for bar in ("1", "0"):
        monkeypatch.setenv("IPCR_CHUNK_BAR", bar)
        monkeypatch.setenv("IPCR_CHUNK_SKIP_INV", "1")
        monkeypatch.delenv("IPCR_CHUNK_HOSTPACK", raising=False)
        big = bytearray(O.bench_dna(20_000_000, 77))
        big[9_000_000:9_000_300] = b"N" * 300
        big[17_500_000:17_500_004] = b"acgt"
        for a in (5_000, 8_388_608 - 90, 16_777_216 - 100, 19_999_800):
            big[a:a + 20] = pairs[0].Forward.encode()
            big[a + 160:a + 180] = O.revcomp(pairs[0].Reverse)
        big = bytes(big)
        got = [p.sig() for p in eng.SimulateCompiledWithScratch("big", big, cp, sc)]
        want = [w.sig() for w in O.simulate_batch(ocfg(cfg), big, opairs(pairs))]
        assert got == want and len(want) >= 4, bar
        assert sc.stats().pattern_set == 1 and sc.stats().hostpack_ms > 0
    monkeypatch.setenv("IPCR_CHUNK_BAR", "1")
    monkeypatch.setenv("IPCR_CHUNK_CLEAN_MODE", "0")           # the knob that puts round 3's behaviour back
    eng.SimulateCompiledWithScratch("s", clean, cp, sc)
    assert sc.stats().pattern_set == 1
    sc.close()
    cp.close()


def test_small_launches_share_a_block_between_waves(hip, monkeypatch):
    """A launch of up to IPCR_JIT_SEG_BLOCKS blocks takes the specialised filter's form in which FOUR waves share a block
    (jit.cpp: jit_source(..., segments): wave s tests the windows that end in its iterations of the rolled loop, runs the
    iteration in front of them only to fill its register window, loads the head quads for the wrap rows itself; the last
    wave's tickets count four waves per block).  The suite's own parity cases on it -- windows across strand and block ends,
    random panels, the cap quirks, long primers (their survivors spill), C2 / C3 / the +N genome, hundreds of tiny records in
    one block, workers sharing a panel -- and the counter says that it is the form that ran."""
    import ctypes
    n_small = hip.lib.lib().ipcr_internal_small_launches
    n_small.restype = ctypes.c_uint64
    monkeypatch.setenv("IPCR_JIT_SEGMENTS", "4")
    before = n_small()
    test_windows_across_strand_and_block_ends(hip, monkeypatch, 0)
    assert n_small() >= before + 3
    mid = n_small()
    for seed in (0, 3):
        test_random_differential(hip, True, seed)
    test_hit_cap_quirks(hip)
    test_halo_case(hip)
    test_long_primers_take_the_specialised_filter(hip, 1)
    test_config_c2_single_pair_k2_tw5(hip)
    test_config_c2_with_reference_n(hip)
    test_config_c3_iupac_k3_circular(hip)
    test_many_tiny_records_in_one_block(hip)
    test_concurrent_workers_share_one_panel(hip)
    test_hit_cap_bounds_device_memory(hip, monkeypatch, 1, 3, 7)      # capped scans repeated over ranges of blocks (block0 > 0)
    assert n_small() > mid + 20
    # launches of more blocks than "small" keep one wave per block
    monkeypatch.setenv("IPCR_JIT_SEG_BLOCKS", "0")
    mid = n_small()
    test_config_c2_single_pair_k2_tw5(hip)
    assert n_small() == mid


def test_resident_genome_scanned_in_rolling_windows(hip, tmp_path):
    """ipcr_scan_genome_chunked: a resident genome scanned the way the pipeline scans it under --chunk-size -- ONE sweep, then
    every rolling window (core/fasta/path_ctx.go:83-179) joined as its own ForEachCompiledProduct call (HitCap, the reset-byte
    rules and the cap-before-window quirk per window; window-local coordinates) -- against the streaming form: the same file
    through fasta.StreamChunks and one ipcr_scan_chunk per chunk, itself checked against the oracle chunk by chunk.  Records
    with runs of N (some windows hold one, some do not), lower case, records shorter than a chunk, an empty record, a header
    without an ID, amplicons inside overlaps and across window starts, dense low-complexity hits under a small cap."""
    from ipcr_amd import fasta
    E, P = hip.engine, hip.primer.Pair
    rng = random.Random(909)
    pairs = hip.primer.AddSelfPairs([P("p", "ACGTTGCATGCAAGCTTA", "GGCCTTAAGGCCATATCG", 0, 0), P("q", "ACGTTG", "TGCAAC", 0, 0)])
    recs = []
    for r, n in enumerate((70_000, 1_500, 0, 33_333, 120_001)):
        s = rand_case(rng, n, with_junk=False) if n else []
        if r in (0, 4):                                     # N runs in some windows only, lower case in others
            for a in (5_000, 41_000):
                if a + 30 < n:
                    s[a:a + 30] = "N" * 30
            s[20_000:20_004] = list("acgt")
        if r == 3:
            s[1000:1400] = list("ACGTTGCA" * 50)            # dense hits for the short pair: the cap bites per window
        for a in range(300, max(n - 400, 0), 3_700):
            plant(rng, s, pairs[0].Forward, a, rng.choice([0, 1]))
            plant(rng, s, O.revcomp(pairs[0].Reverse).decode(), a + rng.randint(60, 250), 0)
        recs.append("".join(s))
    path = tmp_path / "g.fa"
    with open(path, "w") as fh:
        for r, s in enumerate(recs):
            fh.write(">rec%d some text\n" % r)
            for i in range(0, len(s), 61):
                fh.write(s[i:i + 61] + "\n")
            if r == 1:
                fh.write(">\nACGTACGTACGT\n")               # a header without an ID drops its record
    for k, tw, cap, chunk, overlap in ((1, 3, 10000, 9_000, 700), (2, 3, 5, 20_000, 1_000), (1, 0, 0, 9_000, 700), (0, 3, 3, 4_096, 512),
                                       (1, 3, 10000, 0, 0), (1, 3, 10000, 1_000_000, 500)):
        cfg = E.Config(MaxMM=k, TerminalWindow=tw, MinLen=0, MaxLen=400, HitCap=cap, SeedLen=12)
        eng = E.New(cfg)
        cp = eng.CompilePanel(pairs)
        sc = eng.NewSimulationScratch(cp)
        want = []
        for rec in fasta.StreamChunks(str(path), chunk, overlap):
            got_chunk = [(rec.ID,) + p.sig() for p in eng.SimulateCompiledWithScratch(rec.ID, rec.Seq, cp, sc)]
            assert got_chunk == [(rec.ID,) + w.sig() for w in O.simulate_batch(ocfg(cfg), rec.Seq, opairs(pairs))]
            want += got_chunk
        g = E.Genome(sum(len(s) for s in recs) + (1 << 16), max_records=16)
        g.add_fasta(str(path))
        got = [(p.SequenceID,) + p.sig() for p in eng.ScanGenomeChunked(g, cp, sc, chunk, overlap)]
        assert got == want and len(want) >= 20, (k, tw, cap, chunk, overlap)
        g.close()
        sc.close()
        cp.close()


def test_edge_inputs(hip):
    E, P = hip.engine, hip.primer.Pair
    eng = E.New(E.Config(MaxMM=1, TerminalWindow=2, MaxLen=100))
    assert eng.SimulateBatch("s", b"", [P("x", "ACGT", "ACGT")]) == []
    assert eng.SimulateBatch("s", b"ACG", [P("x", "ACGT", "ACGT")]) == []  # shorter than the primer
    assert eng.SimulateBatch("s", b"ACGTACGT", []) == []
    check(hip, E.Config(MaxMM=3, TerminalWindow=9, MaxLen=100), b"ACGTACGTTTACGTAAACGT", [P("x", "ACGT", "ACGT")])  # tw > primer
    check(hip, E.Config(MaxMM=0), b"A" * 9000, [P("x", "AAAA", "TTTT", 0, 20)])  # dense hits, many products
    long_primer = "ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAACCGGATATCGCGATATGGCCAATT" * 2
    seq = "TTT" + long_primer + "GGGG" + O.revcomp(long_primer).decode() + "AAA"
    check(hip, E.Config(MaxMM=2, TerminalWindow=3, MaxLen=1000), seq.encode(), [P("long", long_primer, long_primer)])


def test_errors(hip):
    E, P = hip.engine, hip.primer.Pair
    with pytest.raises(hip.lib.IpcrError) as e:
        E.New(E.Config(MaxMM=-1)).CompilePanel([P("x", "ACGT", "ACGT")])
    assert e.value.status == hip.lib.ERR_INVALID
    with pytest.raises(hip.lib.IpcrError) as e:
        E.New(E.Config()).CompilePanel([P("x", "ACGX", "ACGT")])  # the reference panics (rc.go:27-34)
    assert e.value.status == hip.lib.ERR_PRIMER
    with pytest.raises(hip.lib.IpcrError) as e:
        E.New(E.Config()).CompilePanel([P("x", "A" * 129, "ACGT")])
    assert e.value.status == hip.lib.ERR_UNSUPPORTED


def test_emit_error_aborts(hip):  # core/engine/join_stream_test.go:48-58
    E, P = hip.engine, hip.primer.Pair
    eng = E.New(E.Config(MinLen=6, MaxLen=60))
    cp = eng.CompilePanel([P("join_error", "ACGTAC", "GGTACC", 6, 60)])
    sentinel = RuntimeError("stop joined product stream")
    seen = []

    def emit(p):
        seen.append(p)
        return sentinel

    assert eng.ForEachCompiledProduct("seq", b"TTTACGTACAAAAGGTACCTTT", cp, None, emit) is sentinel
    assert len(seen) == 1


def test_scratch_reuse(hip):  # core/engine/hit_collect_test.go:98-112, compiled_panel_test.go
    E, P = hip.engine, hip.primer.Pair
    eng = E.New(E.Config(MaxMM=0, MinLen=6, MaxLen=60, SeedLen=4))
    cp = eng.CompilePanel([P("x", "ACGTAC", "GGTACC")])
    sc = eng.NewSimulationScratch(cp)
    assert len(eng.SimulateCompiledWithScratch("seq1", b"TTTACGTACAAAAGGTACCTTT", cp, sc)) > 0
    assert eng.SimulateCompiledWithScratch("seq2", b"TTTACGTACAAAACCCCCCCTTT", cp, sc) == []
    big = O.bench_dna(600000, 99)
    eng.SimulateCompiledWithScratch("big", big, cp, sc)   # grows the private tile buffer
    assert len(eng.SimulateCompiledWithScratch("seq1", b"TTTACGTACAAAAGGTACCTTT", cp, sc)) > 0


# ---- tiles, generators -------------------------------------------------------------------------

def test_pack_roundtrip_and_lcg(hip):
    import ctypes as C
    rng = random.Random(5)
    g = hip.engine.Genome(3_000_000, 16)
    recs = []
    for n in (1, 127, 128, 129, 4095, 8192, 8193, 300000, 0):
        s = "".join(rng.choice("ACGTACGTACGTNRacgtn") for _ in range(n)).encode()
        g.add_record("r%d" % n, s)
        recs.append(s)
    for i, s in enumerate(recs):
        assert g.record_len(i) == len(s)
        want = bytes(ch if ch in b"ACGTacgt" else ord("N") for ch in s)
        assert g.read(i, 0, len(s)) == want
        assert bool(g.record_flags(i) & 1) == any(ch not in b"ACGTacgt" for ch in s)
    g.close()


def test_lcg_device_matches_reference_generator(hip):
    torch = pytest.importorskip("torch")
    n = 1_000_003
    t = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    hip.engine.lcg_fill_device(t.data_ptr(), n, 0x5eed1234)
    want = O.bench_dna(n, 0x5eed1234)
    assert bytes(t.cpu().numpy().tobytes()) == want
    hip.engine.lcg_fill_device(t.data_ptr(), 1000, 0x5eed1234, 700_001)  # jump-ahead into the stream
    assert bytes(t[:1000].cpu().numpy().tobytes()) == want[700_001:701_001]


def test_resident_genome_multi_record(hip):
    E, P = hip.engine, hip.primer.Pair
    rng = random.Random(11)
    cfg = E.Config(MaxMM=2, TerminalWindow=3, MaxLen=300, HitCap=10000, SeedLen=12)
    pair = P("p", "ACGTTGCATGCAAGCT", "GGCCTTAAGGCCATAT", 0, 0)
    pairs = hip.primer.AddSelfPairs([pair])
    g = E.Genome(2_000_000, 16)
    seqs = []
    for r in range(7):
        n = rng.choice([5000, 100000, 262144, 262145, 8192 * 3])
        s = rand_case(rng, n, with_junk=(r % 3 == 0))
        for _ in range(5):
            a = rng.randrange(0, n - 300)
            plant(rng, s, pair.Forward, a, rng.choice([0, 1, 2]))
            rc = O.revcomp(pair.Reverse).decode()
            plant(rng, s, rc, a + rng.randint(40, 250), rng.choice([0, 1]))
        s = "".join(s).encode()
        seqs.append(s)
        g.add_record("rec%d" % r, s)
    eng = E.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    got = eng.ScanGenome(g, cp, sc)
    want = []
    op = O.Panel(ocfg(cfg), opairs(pairs))
    for r, s in enumerate(seqs):
        want += [("rec%d" % r,) + w.sig() for w in op.scan(s)]
    assert [(p.SequenceID,) + p.sig() for p in got] == want
    assert len(got) >= 20
    g.close()


# ---- probe -------------------------------------------------------------------------------------

def test_probe_known_answers(hip):  # core/oligo/oligo_test.go:5-21, core/probe/annotate_test.go:5-19
    h = hip.oligo.BestHit("ACGTACGTACGT", "GTAC", 0)
    assert (h.Found, h.Pos, h.MM, h.Strand, h.Site) == (True, 2, 0, "+", "GTAC")
    assert hip.oligo.BestHit("ACGTACGTACGT", "GTGC", 1).Found
    h = hip.oligo.BestHit("AAAGACCC", "GAY", 0)
    assert (h.Found, h.Strand, h.Pos, h.Site) == (True, "+", 3, "GAC")
    a = hip.probe.AnnotateAmplicon("ACGTACGTACGT", "GTAC", 0)
    assert a.Found and a.MM == 0 and a.Pos == 2 and a.Site == "GTAC"
    assert not hip.oligo.BestHit("ACGT", "  ", 0).Found


def test_probe_random_vs_oracle(hip):
    rng = random.Random(3)
    for _ in range(60):
        n = rng.randint(5, 400)
        amp = "".join(rng.choice("ACGTACGTacgtN") for _ in range(n))
        L = rng.randint(3, 25)
        prb = "".join(rng.choice("ACGTACGTACGTRYN") for _ in range(L))
        if rng.random() < 0.5 and n > L + 2:
            p = rng.randrange(0, n - L)
            src = prb if rng.random() < 0.5 else O.revcomp(prb).decode()
            conc = "".join(rng.choice([b for b in "ACGT" if O.base_match(b, ch)]) for ch in src)
            amp = amp[:p] + conc + amp[p + L:]
        k = rng.choice([0, 0, 1, 2])
        w = O.best_hit(amp, prb, k)
        g = hip.oligo.BestHit(amp, prb, k)
        assert (g.Found, g.Strand, g.Pos, g.MM, g.Site) == (w.found, w.strand, w.pos, w.mm, w.site)


# ---- BASELINE.json configurations at test scale (full-scale runs live in bench.py) -------------

def build_planted_genome(hip, rng, nrec, reclen, pairs_to_plant, seed, junk_every=0, wrap=False):
    """LCG records (reference benchDNA) with planted amplicons; returns (Genome, [bytes])."""
    g = hip.engine.Genome(nrec * reclen, nrec)
    seqs = []
    stream = O.bench_dna(nrec * reclen, seed)
    for r in range(nrec):
        s = list(stream[r * reclen:(r + 1) * reclen].decode())
        for t in range(6):
            p = pairs_to_plant[(r + t) % len(pairs_to_plant)]
            a = 1000 + t * ((reclen - 3000) // 6)
            plant(rng, s, p.Forward, a, rng.choice([0, 1, 2]))
            rc = O.revcomp(p.Reverse).decode()
            plant(rng, s, rc, a + 180 - len(rc), rng.choice([0, 1]))
        if wrap:  # reverse site near the record start, forward site near its end: origin-spanning amplicon
            p = pairs_to_plant[0]
            plant(rng, s, p.Forward, reclen - 120, 0)
            rc = O.revcomp(p.Reverse).decode()
            plant(rng, s, rc, 60, 0)
        if junk_every and r % junk_every == 0:
            for _ in range(4):
                q = rng.randrange(reclen - 50)
                s[q:q + rng.randint(1, 40)] = "N" * len(s[q:q + rng.randint(1, 40)])
        b = "".join(s).encode()
        seqs.append(b)
        g.add_record("chr%d" % (r + 1), b)
    return g, seqs


def scan_and_compare(hip, cfg, pairs, g, seqs):
    eng = hip.engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    got = eng.ScanGenome(g, cp, sc)
    op = O.Panel(ocfg(cfg), opairs(pairs))
    want = []
    for r, s in enumerate(seqs):
        want += [("chr%d" % (r + 1),) + w.sig() for w in op.scan(s)]
    assert [(p.SequenceID,) + p.sig() for p in got] == want
    return eng, cp, sc, got


def test_config_c2_single_pair_k2_tw5(hip):
    from ipcr_amd import workloads
    rng = random.Random(21)
    pairs = workloads.c2_pairs()
    g, seqs = build_planted_genome(hip, rng, 6, 2_000_003, pairs[:1], 0x5eed1234, junk_every=0)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    _, _, sc, got = scan_and_compare(hip, cfg, pairs, g, seqs)
    assert sc.stats().kernel_kind == 1 and len(got) >= 20
    g.close()


def test_config_c2_with_reference_n(hip):
    """the '+N' variant of SURVEY 8(d): non-ACGT runs force the reference onto its FindMatches path"""
    from ipcr_amd import workloads
    rng = random.Random(22)
    pairs = workloads.c2_pairs()
    g, seqs = build_planted_genome(hip, rng, 4, 1_500_000, pairs[:1], 0x5eed1235, junk_every=2)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    scan_and_compare(hip, cfg, pairs, g, seqs)
    scan_and_compare(hip, hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=0, SeedLen=12), pairs, g, seqs)
    g.close()


def test_config_c3_iupac_k3_circular(hip):
    from ipcr_amd import workloads
    rng = random.Random(23)
    pairs = workloads.c3_pairs()
    g, seqs = build_planted_genome(hip, rng, 4, 1_000_000, pairs[:1], 0x5eed1236, wrap=True)
    cfg = hip.engine.Config(MaxMM=3, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=True)
    _, _, sc, got = scan_and_compare(hip, cfg, pairs, g, seqs)
    assert sc.stats().kernel_kind == 1
    assert any(p.Start > p.End for p in got), "expected an origin-spanning product"
    g.close()


def test_config_c4_multiplex_panel(hip):
    from ipcr_amd import workloads
    rng = random.Random(24)
    pairs = workloads.c4_pairs(24)  # 24 TSV rows -> 72 pairs, 96 distinct patterns: 8 kernel groups
    g, seqs = build_planted_genome(hip, rng, 3, 300_000, pairs[:24], 0x5eed1237)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng, cp, sc, got = scan_and_compare(hip, cfg, pairs, g, seqs)
    assert sc.stats().kernel_kind == 1 and sc.stats().n_patterns == 96 and len(got) >= 10
    cp2 = eng.CompilePanel(pairs)
    cp2.set_specialize(False)       # same panel on the table-driven filter
    sc2 = eng.NewSimulationScratch(cp2)
    got2 = eng.ScanGenome(g, cp2, sc2)
    assert sc2.stats().kernel_kind == 2 and [p.sig() for p in got2] == [p.sig() for p in got]
    g.close()


def test_config_c5_probe_rescan(hip):
    import ctypes as C
    from ipcr_amd import workloads
    rng = random.Random(25)
    pairs = workloads.c2_pairs()
    g, seqs = build_planted_genome(hip, rng, 3, 800_000, pairs[:1], 0x5eed1238)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=False)
    eng, cp, sc, got = scan_and_compare(hip, cfg, pairs, g, seqs)
    assert got
    amp0 = seqs[got[0].Record][got[0].Start:got[0].End]
    probe = amp0[70:91].decode()                      # 21-mer from the amplicon interior
    probe_mut = probe[:9] + O.different_base(probe[9]) + probe[10:]
    for prb, k in [(probe, 0), (probe_mut, 2), (O.revcomp(probe).decode(), 1), ("ACGTNNRYACGTACGTACGTT", 2)]:
        out = (hip.lib.ProbeHit * len(got))()
        hip.lib.check(hip.lib.lib().ipcr_probe_products(sc._h, g._h, prb.encode(), k, out, len(got)))
        for i, p in enumerate(got):
            amp = seqs[p.Record][p.Start:p.End]
            w = O.best_hit(amp, prb, k)                # core/oligo/oligo.go:19-77 on Product.Seq
            assert (bool(out[i].found), chr(out[i].strand) if out[i].found else "", out[i].pos, out[i].mm) == \
                (w.found, w.strand, w.pos if w.found else 0, w.mm if w.found else 0)
        # the two-halves form: begin returns at once, the same results come out of end; one rescan per scratch at a time
        L = hip.lib.lib()
        out2 = (hip.lib.ProbeHit * len(got))()
        hip.lib.check(L.ipcr_probe_products_begin(sc._h, g._h, prb.encode(), k))
        assert L.ipcr_probe_products_begin(sc._h, g._h, prb.encode(), k) != 0          # not ended yet
        hip.lib.check(L.ipcr_probe_products_end(sc._h, out2, len(got)))
        assert bytes(out2) == bytes(out)
        assert L.ipcr_probe_products_end(sc._h, out2, len(got)) != 0                     # nothing begun
    g.close()


def _probe_tuple(h):
    return (bool(h.found), chr(h.strand) if h.found else "", h.pos if h.found else 0, h.mm if h.found else 0)


def _want_probe(amp, prb, k):
    w = O.best_hit(amp, prb, k)                        # core/oligo/oligo.go:19-77 on Product.Seq
    return (w.found, w.strand, w.pos if w.found else 0, w.mm if w.found else 0)


def test_probe_on_the_chunk_path(hip):
    """ipcr-probe behind the drop-in call (BASELINE C5 as the Go pipeline runs it): ipcr_scan_chunk, then
    ipcr_probe_scratch_products rescans the chunk's products from the tiles that call packed -- against oligo.BestHit
    on the amplicon the pipeline would slice into Product.Seq (internal/pipeline/pipeline.go:80-89,
    internal/visitors/probe.go:18-33).  k = 0 / 1 / 2; probe on '+', on '-', absent, IUPAC; lower-case amplicon bases
    (BestHit upper-cases the amplicon, oligo.go:20); several chunks through one scratch; a panel without products."""
    from ipcr_amd import workloads
    rng = random.Random(41)
    pairs = workloads.c2_pairs()
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = hip.engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    fwd, rcrev = pairs[0].Forward, O.revcomp(pairs[0].Reverse).decode()
    total = 0
    for chunk_no, n in enumerate((300_000, 120_000, 1_000_000, 4_000)):
        s = bytearray(O.bench_dna(n, 0x5eed2000 + chunk_no))
        starts = list(range(1000, n - 400, max(2500, n // 12)))
        for t, a in enumerate(starts):
            f = fwd if t % 3 == 0 else fwd[:7] + O.different_base(fwd[7]) + fwd[8:]
            s[a:a + len(f)] = f.encode()
            s[a + 180 - len(rcrev):a + 180] = rcrev.encode()
            if t % 4 == 1:   # lower-case bases in the amplicon's interior: they match a probe, never a primer
                s[a + 60:a + 100] = bytes(s[a + 60:a + 100]).lower()
            if t % 5 == 2:
                s[a + 75] = ord("N")
        seq = bytes(s)
        got = eng.SimulateCompiledWithScratch("chunk%d" % chunk_no, seq, cp, sc)
        want = O.simulate_batch(ocfg(cfg), seq, opairs(pairs))
        assert [g.sig() for g in got] == [w.sig() for w in want] and len(got) >= len(starts)
        total += len(got)
        a0 = starts[0]
        inner = seq[a0 + 70:a0 + 91].decode().upper()
        mut = inner[:9] + O.different_base(inner[9]) + inner[10:]
        for prb, k in [(inner, 0), (inner, 2), (mut, 0), (mut, 1), (mut, 2), (O.revcomp(inner).decode(), 0),
                       (O.revcomp(mut).decode(), 2), ("ACGTNNRYACGTACGTACGTT", 2), ("GGGGGGGGGGGGGGGGGGGGGGGGG", 0),
                       (inner[:6] + "R" + inner[7:12] + "N" + inner[13:], 1)]:
            out = sc.probe_products(prb, k)
            assert len(out) == len(got)
            for h, p in zip(out, got):
                assert _probe_tuple(h) == _want_probe(seq[p.Start:p.End], prb, k), (chunk_no, prb, k, p)
        # the two halves: begin returns at once, end hands the same records out; one rescan per scratch at a time
        import ctypes as C
        L = hip.lib.lib()
        ref = sc.probe_products(inner, 2)
        out2 = (hip.lib.ProbeHit * len(got))()
        hip.lib.check(L.ipcr_probe_scratch_products_begin(sc._h, inner.encode(), 2))
        assert L.ipcr_probe_scratch_products_begin(sc._h, inner.encode(), 2) != 0
        hip.lib.check(L.ipcr_probe_products_end(sc._h, out2, len(got)))
        assert [_probe_tuple(out2[i]) for i in range(len(got))] == [_probe_tuple(h) for h in ref]
        assert L.ipcr_probe_scratch_products(sc._h, inner.encode(), 2, out2, len(got) + 1) != 0   # n_out must match
    assert total >= 30
    # a chunk without products, and an invalid probe
    assert eng.SimulateCompiledWithScratch("empty", b"ACGT" * 1000, cp, sc) == [] and sc.probe_products("ACGTACGT", 1) == []
    with pytest.raises(hip.lib.IpcrError):
        sc.probe_products("ACGU", 0)
    # a scratch whose last scan was a resident genome's has no chunk tiles to read
    g = hip.engine.Genome(100_000, 1)
    g.add_record("r", O.bench_dna(50_000, 7))
    eng.ScanGenome(g, cp, sc)
    with pytest.raises(hip.lib.IpcrError):
        sc.probe_products("ACGTACGT", 1)
    g.close()
    cp.close()


def test_probe_on_the_chunk_path_wrap_product(hip):
    """--circular: the record goes through ipcr_scan_chunk whole (chunking is off, internal/runutil/runutil.go:46-49) and an
    origin-spanning product's amplicon is record[start:] ++ record[:end] (pipeline.go:82-84): probes that lie across the
    junction, on either strand, and one that lies before it."""
    P = hip.primer.Pair
    fwd, rev = "ACGTTGCATGCAAGCTTGCA", "GGCCTTAAGGCCATATCCGG"
    n = 60_000
    s = bytearray(O.bench_dna(n, 0x77aa))
    s[n - 100:n - 100 + len(fwd)] = fwd.encode()
    rc = O.revcomp(rev)
    s[80 - len(rc):80] = rc
    s[20_000:20_000 + len(fwd)] = fwd.encode()            # and an ordinary product
    s[20_000 + 300 - len(rc):20_000 + 300] = rc
    seq = bytes(s)
    pairs = [P("w", fwd, rev, 0, 0)]
    cfg = hip.engine.Config(MaxMM=1, TerminalWindow=3, MaxLen=1000, HitCap=10000, SeedLen=12, Circular=True)
    eng = hip.engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    got = eng.SimulateCompiledWithScratch("plasmid", seq, cp, sc)
    want = O.simulate_batch(ocfg(cfg), seq, opairs(pairs))
    assert [g.sig() for g in got] == [w.sig() for w in want]
    wraps = [p for p in got if p.Start > p.End]
    assert wraps and any(p.Start <= p.End for p in got)
    amp_of = lambda p: seq[p.Start:p.End] if p.Start <= p.End else seq[p.Start:] + seq[:p.End]
    w0 = amp_of(wraps[0])
    junction = n - wraps[0].Start
    across = w0[junction - 10:junction + 11].decode()     # 21-mer over the origin
    before = w0[30:51].decode()
    for prb, k in [(across, 0), (O.revcomp(across).decode(), 0), (across[:5] + O.different_base(across[5]) + across[6:], 1),
                   (before, 0), ("TTTTTTTTTTTTTTTTTTTTTT", 0)]:
        out = sc.probe_products(prb, k)
        for h, p in zip(out, got):
            assert _probe_tuple(h) == _want_probe(amp_of(p), prb, k), (prb, k, p)
    assert _probe_tuple(sc.probe_products(across, 0)[got.index(wraps[0])])[0]
    cp.close()


def test_probe_amplicons_beyond_the_lds_stage(hip):
    """--max-length above the probe kernel's LDS stage (16 384 bases): the batched rescan falls back to gather + rescan
    from device memory (two launches, the same tagged hand-over), on the chunk path and over a resident genome; a batch
    that mixes a long amplicon with short ones goes the same way as a whole."""
    P = hip.primer.Pair
    fwd, rev = "ACGTTGCATGCAAGCTTGCA", "GGCCTTAAGGCCATATCCGG"
    rc = O.revcomp(rev)
    n = 120_000
    s = bytearray(O.bench_dna(n, 0x1ab5))
    s[1000:1020] = fwd.encode()
    s[1000 + 30_000 - 20:1000 + 30_000] = rc          # a 30 000-base product
    s[60_000:60_020] = fwd.encode()
    s[60_300 - 20:60_300] = rc                         # and a 300-base one
    seq = bytes(s)
    pairs = [P("long", fwd, rev, 0, 0)]
    cfg = hip.engine.Config(MaxMM=1, TerminalWindow=3, MaxLen=40_000, HitCap=10000, SeedLen=12)
    eng = hip.engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    got = eng.SimulateCompiledWithScratch("r", seq, cp, sc)
    want = O.simulate_batch(ocfg(cfg), seq, opairs(pairs))
    assert [g.sig() for g in got] == [w.sig() for w in want]
    assert any(p.Length == 30_000 for p in got) and any(p.Length == 300 for p in got)
    deep = seq[1000 + 20_000:1000 + 20_021].decode()   # a probe site 20 000 bases into the long amplicon
    for prb, k in [(deep, 0), (O.revcomp(deep).decode(), 1), (seq[60_100:60_121].decode(), 0), ("TTTTTTTTTTTTTTTTTTTTTTTT", 0)]:
        out = sc.probe_products(prb, k)
        for h, p in zip(out, got):
            assert _probe_tuple(h) == _want_probe(seq[p.Start:p.End], prb, k), (prb, k, p)
    g = hip.engine.Genome(n + 8192, 1)
    g.add_record("r", seq)
    got_g = eng.ScanGenome(g, cp, sc)
    assert [p.sig() for p in got_g] == [w.sig() for w in want]
    out = (hip.lib.ProbeHit * len(got_g))()
    hip.lib.check(hip.lib.lib().ipcr_probe_products(sc._h, g._h, deep.encode(), 0, out, len(got_g)))
    for i, p in enumerate(got_g):
        assert _probe_tuple(out[i]) == _want_probe(seq[p.Start:p.End], deep, 0)
    g.close()
    cp.close()


def test_probe_best_hit_from_many_threads(hip):
    """ipcr_probe_best_hit is what a collector calls per product next to the workers' sweeps: no allocation, no
    null-stream work, one pinned block and stream per concurrent caller.  Eight threads, amplicons of 5..20 000 bases
    (beyond the kernel's LDS stage: read from pinned memory directly), against the oracle."""
    import threading
    errs = []

    def worker(seed):
        rng = random.Random(seed)
        try:
            for it in range(40):
                n = rng.choice([5, 40, 180, 700, 2000, 2000, 6000]) if it else 20_000
                amp = "".join(rng.choice("ACGTACGTACGTacgtN") for _ in range(n))
                L = rng.randint(3, 30)
                prb = "".join(rng.choice("ACGTACGTACGTRYN") for _ in range(L))
                if rng.random() < 0.6 and n > L + 2:
                    q = rng.randrange(0, n - L)
                    src = prb if rng.random() < 0.5 else O.revcomp(prb).decode()
                    conc = "".join(rng.choice([b for b in "ACGT" if O.base_match(b, ch)]) for ch in src)
                    amp = amp[:q] + conc + amp[q + L:]
                k = rng.choice([0, 0, 1, 2])
                w = O.best_hit(amp, prb, k)
                g = hip.oligo.BestHit(amp, prb, k)
                if (g.Found, g.Strand, g.Pos, g.MM, g.Site) != (w.found, w.strand, w.pos, w.mm, w.site):
                    errs.append((seed, it, n, prb, k, g, w))
        except Exception as e:  # noqa: BLE001
            errs.append((seed, repr(e)))

    ts = [threading.Thread(target=worker, args=(100 + i,)) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:3]


# ---- seed-index filter (large panels) ------------------------------------------------------------

@pytest.fixture
def force_index(monkeypatch):
    monkeypatch.setenv("IPCR_FORCE_INDEX", "1")


@pytest.mark.parametrize("seed", range(4))
def test_index_filter_random(hip, force_index, monkeypatch, seed):
    """same differential test as above, every panel pushed through the seed-index filter
    (IUPAC primers take its table-driven side path); seeds 1 and 2 with one and two base steps per queue entry
    instead of as many as the entry layout allows (jit.cpp: jit_index_source)"""
    if seed in (1, 2):
        monkeypatch.setenv("IPCR_INDEX_STEPS_PER_ENTRY", str(seed))
    rng = random.Random(4321 + seed)
    E, P = hip.engine, hip.primer.Pair
    kinds = set()
    for it in range(8):
        n = rng.choice([300, 5000, 70000])
        seq = rand_case(rng, n, with_junk=rng.random() < 0.5)
        lmin = 6 if n <= 300 else (9 if n <= 5000 else 14)
        pairs = []
        for i in range(rng.randint(1, 5)):
            def mk():
                L = rng.randint(lmin, 30)
                s = [rng.choice("ACGT") for _ in range(L)]
                if rng.random() < 0.2:
                    s[rng.randrange(L)] = rng.choice("RYSWKMBDHVN")
                return "".join(s)
            pairs.append(P("p%d" % i, mk(), mk(), rng.choice([0, 0, 20]), rng.choice([0, 0, 400])))
        for p in pairs:
            for _ in range(rng.randint(1, 3)):
                a = rng.randrange(0, max(1, n - 200))
                ln = rng.randint(len(p.Forward) + len(p.Reverse), 150)
                if a + ln > n:
                    continue
                plant(rng, seq, p.Forward, a, rng.choice([0, 0, 1, 2]))
                rc = O.revcomp(p.Reverse).decode()
                plant(rng, seq, rc, a + ln - len(rc), rng.choice([0, 0, 1]))
        cfg = E.Config(MaxMM=rng.choice([0, 1, 2, 3] if n <= 5000 else [0, 1, 2]), TerminalWindow=rng.choice([0, 1, 3, 5, 9]),
                       MinLen=rng.choice([0, 10]), MaxLen=rng.choice([0, 200, 2000]),
                       HitCap=rng.choice([0, 0, 3, 10000]), SeedLen=rng.choice([0, 12, -1]), Circular=rng.random() < 0.3)
        if rng.random() < 0.5:
            pairs = hip.primer.AddSelfPairs(pairs)
        eng = E.New(cfg)
        cp = eng.CompilePanel(pairs)
        sc = eng.NewSimulationScratch(cp)
        got = eng.SimulateCompiledWithScratch("seq", "".join(seq).encode(), cp, sc)
        kinds.add(sc.stats().kernel_kind)
        want = O.simulate_batch(ocfg(cfg), "".join(seq).encode(), opairs(pairs))
        assert [g.sig() for g in got] == [w.sig() for w in want], (cfg, pairs)
    assert 3 in kinds


@pytest.mark.parametrize("half_bases,two_step", [(0, 0), (1, 0), (0, 1)])
def test_config_c4_large_panel_index(hip, monkeypatch, half_bases, two_step):
    """256 TSV rows -> 768 pairs / 1024 distinct patterns: seed-index filter, vs the oracle; with 16-bit keys, with the
    17-bit keys (a block takes one bit of a spare sixth base: host.cpp build_index, the default), and with the tables
    that serve two base steps per lookup (jit.cpp, IPCR_INDEX_TWO_STEP: measured slower, kept as a knob)"""
    from ipcr_amd import workloads
    monkeypatch.setenv("IPCR_INDEX_HALF_BASES", str(half_bases))
    monkeypatch.setenv("IPCR_INDEX_TWO_STEP", str(two_step))
    rng = random.Random(26)
    pairs = workloads.c4_pairs(256)
    g, seqs = build_planted_genome(hip, rng, 3, 400_000, pairs[:256], 0x5eed1239)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    _, cp, sc, got = scan_and_compare(hip, cfg, pairs, g, seqs)
    assert sc.stats().kernel_kind == 3 and sc.stats().n_patterns == 1024 and len(got) >= 10
    g.close()


@pytest.mark.parametrize("k,tw", [(2, 3), (1, 5), (3, 3), (2, 1)])
def test_index_split_keys_for_orientations_scanned_without_their_window(hip, monkeypatch, k, tw):
    """Every ipcr_scan_chunk call (and every record of a genome that holds an N) scans the rc orientations WITHOUT their 5'
    window -- the reference caps them before the window filter (core/engine/compiled.go:249-256) -- and the seed index files
    such a pattern under two families of keys: window exact + one of k + 1 blocks, or a mismatch in the window + one of k
    longer blocks (host.cpp: build_index, SPLIT).  A 256-row panel through the chunk path, records with and without N,
    sites with their mismatches inside and outside the window, against the oracle -- with the split and with the old
    keys (IPCR_INDEX_SPLIT=0): the same products either way, and the split really is in the generated kernel."""
    from ipcr_amd import workloads
    rng = random.Random(1000 * k + tw)
    pairs = workloads.c4_pairs(256)
    rows = pairs[:256]
    cfg = hip.engine.Config(MaxMM=k, TerminalWindow=tw, MaxLen=2000, HitCap=10000, SeedLen=12)
    seqs = []
    for with_n in (True, False):
        s = bytearray(O.bench_dna(300_000, 0x5eed3000 + k + 16 * tw + int(with_n)))
        for t in range(60):
            p = rows[rng.randrange(len(rows))]
            a = 2000 + t * 4800 + rng.randrange(500)
            f = list(p.Forward)
            rc = list(O.revcomp(p.Reverse).decode())
            for _ in range(rng.choice([0, 0, 1, k])):        # mismatches anywhere in the forward site: inside its 3' window too
                j = rng.randrange(len(f)); f[j] = O.different_base(f[j])
            for _ in range(rng.choice([0, 1, k])):           # ... and in the rc site: inside its 5' window too (raw matches the host filters)
                j = rng.randrange(len(rc)); rc[j] = O.different_base(rc[j])
            s[a:a + len(f)] = "".join(f).encode()
            s[a + 180 - len(rc):a + 180] = "".join(rc).encode()
        if with_n:
            for _ in range(12):
                q = rng.randrange(len(s) - 50); s[q:q + rng.randint(1, 30)] = b"N" * 30
        seqs.append(bytes(s[:300_000]))
    results = {}
    for split in ("1", "0"):
        monkeypatch.setenv("IPCR_INDEX_SPLIT", split)
        eng = hip.engine.New(cfg)
        cp = eng.CompilePanel(pairs)
        sc = eng.NewSimulationScratch(cp)
        head = cp.filter_source(3).splitlines()[0]
        nshapes = int(head.split("),")[1].split("key shapes")[0]) if ")," in head else int(head.split(", ")[1].split(" key shapes")[0])
        out = []
        for seq in seqs:
            got = eng.SimulateCompiledWithScratch("chunk", seq, cp, sc)
            assert sc.stats().kernel_kind == 3
            want = O.simulate_batch(ocfg(cfg), seq, opairs(pairs))
            assert [g.sig() for g in got] == [w.sig() for w in want] and len(want) >= 10, (k, tw, split)
            out.append([g.sig() for g in got])
        results[split] = (nshapes, out)
        sc.close(); cp.close()
    assert results["1"][1] == results["0"][1]
    # right group k + 1 shapes; rc group: (k + 1) + k with the split, k + 1 without
    assert results["1"][0] == 3 * k + 2 and results["0"][0] == 2 * (k + 1), results


def test_buffer_regrowth_on_dense_hits(hip):
    """millions of hits: the device hit buffer (1 Mi records) and candidate queue must regrow and the
    scan re-run transparently; HitCap still keeps only the first matches per orientation"""
    E, P = hip.engine, hip.primer.Pair
    n = 2_500_000
    seq = b"A" * n
    cfg = E.Config(MaxMM=0, TerminalWindow=3, MaxLen=40, HitCap=4)
    eng = E.New(cfg)
    cp = eng.CompilePanel([P("polyA", "AAAAAAAAAAAA", "TTTTTTTTTTTT")])
    sc = eng.NewSimulationScratch(cp)
    got = eng.SimulateCompiledWithScratch("s", seq, cp, sc)
    assert sc.stats().hits >= 2 * (n - 11)          # A and rc(B)=A...A both match everywhere
    # dense survivors overflow the waves' own verify lists: the spilled words went through the queue and
    # the stand-alone verifier (sparse scans never launch it, see test_pipelined_begin_end)
    assert sc.stats().kernel_kind == 1 and sc.stats().verify_ms > 0
    want = O.simulate_batch(ocfg(cfg), seq[:200000], opairs(cp.Pairs))  # capped lists: a prefix decides
    assert [g.sig() for g in got] == [w.sig() for w in want] and len(got) > 0
    # and the scratch is still healthy afterwards
    assert len(eng.SimulateCompiledWithScratch("seq1", b"AAAAAAAAAAAAGAAAAAAAAAAAA", cp, sc)) >= 1


def test_concurrent_workers_share_one_panel(hip):
    """the pipeline's threading contract (internal/pipeline/pipeline.go:60-125): CompilePanel once,
    one scratch per worker, ForEachCompiledProduct concurrently from several workers"""
    import threading
    E, P = hip.engine, hip.primer.Pair
    cfg = E.Config(MaxMM=1, TerminalWindow=3, MaxLen=400, HitCap=10000, SeedLen=12)
    pairs = hip.primer.AddSelfPairs([P("p", "ACGTTGCATGCAAGCT", "GGCCTTAAGGCCATAT")])
    eng = E.New(cfg)
    cp = eng.CompilePanel(pairs)
    rng = random.Random(77)
    jobs = []
    for j in range(24):
        n = rng.choice([3000, 50000, 300000])
        s = rand_case(rng, n, with_junk=(j % 4 == 0))
        for _ in range(4):
            a = rng.randrange(0, n - 400)
            plant(rng, s, pairs[0].Forward, a, rng.choice([0, 1]))
            plant(rng, s, O.revcomp(pairs[0].Reverse).decode(), a + rng.randint(40, 300), 0)
        jobs.append("".join(s).encode())
    want = [[w.sig() for w in O.simulate_batch(ocfg(cfg), s, opairs(pairs))] for s in jobs]
    got = [None] * len(jobs)
    errors = []

    def worker(wid):
        try:
            sc = eng.NewSimulationScratch(cp)
            for j in range(wid, len(jobs), 4):
                got[j] = [p.sig() for p in eng.SimulateCompiledWithScratch("job%d" % j, jobs[j], cp, sc)]
            sc.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and got == want
    # several scratches alive = several workers: ipcr_scan_chunk stages the caller's bytes through two pinned 8 MiB
    # slices per scratch (CPU copy of slice i+1 under the DMA of slice i); a 20 Mb record takes three slices
    s1, s2 = eng.NewSimulationScratch(cp), eng.NewSimulationScratch(cp)
    big = bytearray(O.bench_dna(20_000_000, 4242))
    big[5_000_000:5_000_003] = b"NNN"
    for a in (100, 8_388_500, 16_777_100, 19_999_000):     # amplicons across both slice boundaries and at the ends
        big[a:a + 16] = pairs[0].Forward.encode()
        big[a + 200:a + 216] = O.revcomp(pairs[0].Reverse)
    big = bytes(big)
    want_big = [w.sig() for w in O.simulate_batch(ocfg(cfg), big, opairs(pairs))]
    for sc in (s1, s2, s1):
        assert [p.sig() for p in eng.SimulateCompiledWithScratch("big", big, cp, sc)] == want_big and len(want_big) >= 4
    # a chunk of up to 8 MiB goes in two halves (the split is rounded up to 4 KiB): lengths around the roundings, an
    # amplicon across the split, and the degenerate ones
    for n in (0, 1, 15, 16, 4095, 4096, 4097, 8191, 8193, 100_001, 8_388_608):
        seq = bytearray(O.bench_dna(n, 99 + n)) if n else bytearray()
        half = (((n + 1) // 2) + 4095) & ~4095
        if n >= 8193:
            a = min(half, n - 300) - 100
            seq[a:a + 16] = pairs[0].Forward.encode()
            seq[a + 200:a + 216] = O.revcomp(pairs[0].Reverse)
        seq = bytes(seq)
        want_n = [w.sig() for w in O.simulate_batch(ocfg(cfg), seq, opairs(pairs))]
        assert [p.sig() for p in eng.SimulateCompiledWithScratch("n%d" % n, seq, cp, s2)] == want_n, n
        assert n < 8193 or len(want_n) >= 1
    s1.close()
    s2.close()


@pytest.mark.parametrize("npairs,count,junk", [(6, 3, 0), (40, 2, 2), (40, 4, 0)])
def test_pattern_shards_scan_and_join(hip, npairs, count, junk):
    """ipcr_panel_set_shard on the device: `count` panel objects scan the same resident genome with slices of the
    distinct-pattern list (specialised kernels for the small panel, the seed index for the 40-row one); the
    concatenated hit lists joined with the full panel are the unsharded scan's products, which equal the oracle's"""
    import numpy as np
    from ipcr_amd import dist, workloads
    rng = random.Random(700 + npairs + count)
    pairs = workloads.c4_pairs(npairs)
    g, seqs = build_planted_genome(hip, rng, 4, 400_003, pairs[:npairs], 0x5eed7777, junk_every=junk)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng, full, sc, want = scan_and_compare(hip, cfg, pairs, g, seqs)
    parts, kinds = [], set()
    for i in range(count):
        sh = eng.CompilePanel(pairs)
        sh.set_shard(i, count)
        ssc = eng.NewSimulationScratch(sh)
        eng.ScanGenomeHits(g, sh, ssc)
        kinds.add(ssc.stats().kernel_kind)
        part = dist.hits_from_scratch(ssc)
        assert set(int(x) & 0x7FFFFFFF for x in part["pattern"]) <= set(sh.scanned_patterns(0)) | set(sh.scanned_patterns(1))
        parts.append(part)
        ssc.close()
        sh.close()
    host = hip.engine.SimulationScratch(full, host_only=True)
    lens = [g.record_len(r) for r in range(g.num_records)]
    flags = [g.record_flags(r) for r in range(g.num_records)]
    got = eng.JoinHits(full, host, np.concatenate(parts[::-1]), lens, flags, g.ids)
    assert [p.sig() for p in got] == [p.sig() for p in want] and len(want) >= 10
    assert kinds <= {1, 3}
    g.close()


def test_index_large_iupac_panel(hip):
    """a 150-row panel where a third of the primers carry IUPAC codes: the seed index expands them
    into concrete keys and checks them with per-base masks; vs the oracle"""
    rng = random.Random(27)
    P = hip.primer.Pair
    rows = []
    for i in range(150):
        def mk():
            L = rng.randint(18, 25)
            s = [rng.choice("ACGT") for _ in range(L)]
            if rng.random() < 0.33:
                for _ in range(rng.randint(1, 3)):
                    s[rng.randrange(L)] = rng.choice("RYSWKMBDHVN")
            return "".join(s)
        rows.append(P("row%03d" % i, mk(), mk(), 0, 0))
    pairs = hip.primer.AddSelfPairsUnique(rows)
    g, seqs = build_planted_genome(hip, rng, 2, 300_000, rows, 0x5eed123a, junk_every=2)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    _, cp, sc, got = scan_and_compare(hip, cfg, pairs, g, seqs)
    assert sc.stats().kernel_kind == 3 and sc.stats().n_patterns >= 590 and len(got) >= 4
    g.close()


def test_k4_medium_panel_is_specialised_in_groups(hip):
    """k = 4 is beyond the seed index (five pigeonhole blocks of three bases key nothing): a 30-row panel (120
    patterns, 10 groups) must still not fall to the table-driven filter -- the specialised filter takes it in up to
    32 groups, one sweep each; vs the oracle"""
    rng = random.Random(31)
    P = hip.primer.Pair
    rows = [P("row%03d" % i, "".join(rng.choice("ACGT") for _ in range(rng.randint(18, 24))),
              "".join(rng.choice("ACGT") for _ in range(rng.randint(18, 24))), 0, 0) for i in range(30)]
    g, seqs = build_planted_genome(hip, rng, 2, 150_000, rows, 0x5eed1240)
    cfg = hip.engine.Config(MaxMM=4, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    _, cp, sc, got = scan_and_compare(hip, cfg, rows, g, seqs)
    assert sc.stats().n_patterns == 120 and sc.stats().kernel_kind == 1 and len(got) >= 4
    g.close()


@pytest.mark.parametrize("stack", [1, 0, 2])
def test_index_drain_under_chains_and_crowded_rounds(hip, force_index, monkeypatch, stack):
    """(stack = 1: the drain takes the newest 64 entries per round and leaves fewer than 64 for the next drain of the
    unit, jit.cpp: stack_drain; 0: the front-to-back form it replaced, still behind IPCR_INDEX_STACK_DRAIN=0, with the
    units taken at their start; 2: the stack drain with one unit counter for the chip, IPCR_INDEX_XCD=0)
    worst case for the seed-index drain: three families of 16 primers that differ only in two bases of one block
    (every key of the other blocks is shared by the whole family: entry chains of 16; an exact site is filed under
    all three shapes, and with k = 2 every site matches all 16 members) on a sequence
    of period 128 = one strand, so that all 64 lanes of a wave hit in the same base step and every lane hands entries
    back to the queue in every round; HitCap 0 keeps every match.  vs the oracle"""
    rng = random.Random(77)
    E, P = hip.engine, hip.primer.Pair
    left, right = "ACGTTGCA", "GGATCCTAAC"          # 8 + 2 + 10 = 20 nt; the last 3 bases are protected at k = 2
    fam = [left + a + b + right for a in "ACGT" for b in "ACGT"]
    mids = [x[8:10] for x in fam]
    rev = "TTGACCGTAGGCATTCAGGA"
    pairs = [P("f%02d" % i, f, rev, 0, 0) for i, f in enumerate(fam)]
    # two more families: the middle bases sit in another block, so other shapes carry the chains
    for j, (l2, r2) in enumerate([("CATG", "ACCGTTAGCATCGG"), ("TGCATGCATGCAAT", "GTCA")]):
        pairs += [P("g%d_%02d" % (j, i), l2 + m + r2, rev, 0, 0) for i, m in enumerate(mids)]
    unit = list("".join(rng.choice("ACGT") for _ in range(128)))
    unit[10:30] = fam[5]
    unit[40:60] = list(pairs[20].Forward)
    unit[70:90] = list(pairs[40].Forward)
    rc = O.revcomp(rev.encode()).decode()
    unit[100:120] = rc
    seq = ("".join(unit) * 600).encode()            # 76 800 bases: 600 strands, each with the same sites
    cfg = E.Config(MaxMM=2, TerminalWindow=3, MinLen=0, MaxLen=120, HitCap=0, SeedLen=12)
    monkeypatch.setenv("IPCR_INDEX_STACK_DRAIN", str(min(stack, 1)))   # read when the kernel's source is generated (first scan)
    if stack == 2:   # the unit hand-out it replaced too: one counter for the chip, a unit's loads at its start
        monkeypatch.setenv("IPCR_INDEX_XCD", "0")
    monkeypatch.setenv("IPCR_INDEX_AHEAD", "0" if stack == 0 else "1")   # (no effect without the per-XCD counters)
    eng = E.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    got = eng.SimulateCompiledWithScratch("seq", seq, cp, sc)
    assert sc.stats().kernel_kind == 3
    want = O.simulate_batch(ocfg(cfg), seq, opairs(pairs))
    assert len(want) >= 600 * 16 * 3
    assert [g.sig() for g in got] == [w.sig() for w in want]


def test_index_panel_leftovers_take_specialised_filters(hip):
    """a 120-row panel (seed index) in which four rows have primers of 36-40 nt and two have four N in what would be
    their keys: the index cannot key those patterns, and they must not fall to the table-driven kernel (~4 ms per
    pattern and 3 Gb) -- spill-only specialised filters take them, their survivors join the index's in the candidate
    queue; vs the oracle"""
    rng = random.Random(29)
    P = hip.primer.Pair
    def mk(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    rows = [P("row%03d" % i, mk(rng.randint(18, 25)), mk(rng.randint(18, 25)), 0, 0) for i in range(114)]
    rows += [P("long%d" % i, mk(36 + i), mk(40 - i), 0, 0) for i in range(4)]
    for i in range(2):   # N at four of the five bases of the block next to the protected end: 256 expansions of that key
        f = list(mk(20)); f[13:17] = "NNNN"
        rows.append(P("deg%d" % i, "".join(f), mk(20), 0, 0))
    plant_rows = rows[:6] + rows[114:]
    g, seqs = build_planted_genome(hip, rng, 2, 300_000, plant_rows, 0x5eed1241)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    _, cp, sc, got = scan_and_compare(hip, cfg, rows, g, seqs)
    st = sc.stats()
    assert st.kernel_kind == 3 and st.leftover_patterns >= 18 and st.leftover_kernels >= 2 and len(got) >= 8
    g.close()


def test_products_equal_the_definition(hip):
    """the HIP path against the product list stated as a definition (tests/test_join_definition.py: plain-Python sets
    and a sort key, no code shared with the oracle or with host.cpp's join) -- the third side of the triangle"""
    from test_join_definition import products_by_definition, IUPAC, rc
    rng = random.Random(9300)
    E, P = hip.engine, hip.primer.Pair
    total = 0
    for case in range(30):
        n = rng.choice([40, 200, 3000])
        seq = [rng.choice("ACGT") for _ in range(n)]
        k, tw = rng.choice([0, 1, 2]), rng.choice([0, 2, 3])
        rows = []
        for i in range(rng.randint(1, 3)):
            f = "".join(rng.choice("ACGT") for _ in range(rng.randint(6, 12)))
            r = "".join(rng.choice("ACGTRYN") if rng.random() < 0.1 else rng.choice("ACGT") for _ in range(rng.randint(6, 12)))
            for _ in range(2):
                a = rng.randrange(0, max(1, n - 40))
                seq[a:a + len(f)] = list(f)
                rr = [rng.choice(IUPAC[c]) for c in rc(r)]
                seq[a + 20:a + 20 + len(rr)] = rr
            rows.append(("p%d" % i, f, r, rng.choice([0, 0, 10]), rng.choice([0, 0, 60])))
        cmin, cmax, circular = rng.choice([0, 8]), rng.choice([0, 100]), rng.random() < 0.4
        s = "".join(seq[:n])
        want = []
        for pid, f, r, pmin, pmax in rows:
            want += products_by_definition(s, pid, f, r, k, tw, pmin or cmin, pmax or cmax, circular)
        cfg = E.Config(MaxMM=k, TerminalWindow=tw, MinLen=cmin, MaxLen=cmax, HitCap=0, SeedLen=rng.choice([0, 6, -1]), Circular=circular)
        eng = E.New(cfg)
        cp = eng.CompilePanel([P(*r) for r in rows])
        sc = eng.NewSimulationScratch(cp)
        got = [p.sig() for p in eng.SimulateCompiledWithScratch("s", s.encode(), cp, sc)]
        assert got == want, (s, rows, k, tw, cmin, cmax, circular)
        total += len(want)
        sc.close()
        cp.close()
    assert total > 60


def test_need_sites(hip):  # core/engine/engine.go:175-183 (FwdSite / RevSite for pretty text)
    E, P = hip.engine, hip.primer.Pair
    seq = b"TTTTCGTACAAAAGGTACCTTT"
    eng = E.New(E.Config(MaxMM=1, TerminalWindow=3, MinLen=1, MaxLen=100, SeedLen=12, NeedSites=True))
    got = eng.SimulateBatch("seq", seq, [P("x", "ACGTAC", "GGTACC")])
    by = {(p.Type, p.Start): p for p in got}
    f = by[("forward", 3)]
    assert (f.FwdPrimer, f.RevPrimer, f.FwdSite, f.RevSite) == ("ACGTAC", "GGTACC", "TCGTAC", "GGTACC")
    r = by[("revcomp", 13)]
    assert (r.FwdPrimer, r.RevPrimer, r.FwdSite, r.RevSite) == ("GGTACC", "ACGTAC", "GGTACC", "AGGTAC")
    plain = E.New(E.Config(MaxMM=1, TerminalWindow=3, MinLen=1, MaxLen=100)).SimulateBatch("seq", seq, [P("x", "ACGTAC", "GGTACC")])
    assert all(p.FwdSite == "" and p.RevSite == "" for p in plain)


def test_pipelined_begin_end(hip):
    """ipcr_scan_genome_begin / _end on two alternating scratches (the bench's pipelining) give the
    same products as the one-call scan, pass after pass"""
    from ipcr_amd import workloads
    rng = random.Random(31)
    pairs = workloads.c2_pairs()
    g, seqs = build_planted_genome(hip, rng, 3, 700_000, pairs[:1], 0x5eed123b, junk_every=3)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = hip.engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    a, b = eng.NewSimulationScratch(cp), eng.NewSimulationScratch(cp)
    want = [p.sig() for p in eng.ScanGenome(g, cp, a)]
    assert want
    st = a.stats()  # specialised filter, survivors verified by the filter's own waves: one kernel per scan
    assert st.kernel_kind == 1 and st.verify_ms == 0 and st.hits > 0 and st.candidates >= st.hits
    scs = [a, b]
    eng.ScanGenomeBegin(g, cp, scs[0])
    for i in range(6):
        if i + 1 < 6:
            scs[(i + 1) & 1].chain_after(scs[i & 1])
            eng.ScanGenomeBegin(g, cp, scs[(i + 1) & 1])
        n = eng.ScanGenomeEndCount(g, cp, scs[i & 1])
        assert n == len(want) and [p.sig() for p in scs[i & 1].products(g.ids)] == want
    with pytest.raises(hip.lib.IpcrError):  # nothing in flight any more
        eng.ScanGenomeEndCount(g, cp, a)
    # many chained passes over three scratches: every pass must hand over complete results (counters, hit
    # records and the sequence word reach pinned memory from different waves)
    scs = [a, b, eng.NewSimulationScratch(cp)]
    nhits = a.stats().hits
    eng.ScanGenomeBegin(g, cp, scs[0])
    for i in range(600):
        if i + 1 < 600:
            scs[(i + 1) % 3].chain_after(scs[i % 3])
            eng.ScanGenomeBegin(g, cp, scs[(i + 1) % 3])
        assert eng.ScanGenomeEndCount(g, cp, scs[i % 3]) == len(want) and scs[i % 3].stats().hits == nhits, i
    g.close()


def test_device_hit_exchange_one_rank_rccl(hip, monkeypatch):
    """ipcr_scratch_device_hits + HitExchanger.start_scratch: the all-gather reads the scratch's device hit
    buffer through a zero-copy torch view (one-rank RCCL group: the collective, the header parsing and the
    two-slot bookkeeping are the ones a multi-GPU job runs)"""
    import numpy as np
    import torch
    import torch.distributed as tdist
    from ipcr_amd import dist, workloads
    monkeypatch.setenv("IPCR_EXCHANGE_SELFTEST", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29531")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    rng = random.Random(77)
    pairs = workloads.c2_pairs()
    g, _ = build_planted_genome(hip, rng, 3, 600_000, pairs[:1], 0x5eed4444, junk_every=2)
    eng = hip.engine.New(hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12))
    cp = eng.CompilePanel(pairs)
    scs = [eng.NewSimulationScratch(cp) for _ in range(3)]
    started_here = not tdist.is_initialized()
    dist.init_process_group("nccl")
    try:
        dev = torch.device("cuda", 0)
        x = dist.HitExchanger(device=dev)
        want_products = [p.sig() for p in eng.ScanGenome(g, cp, scs[0])]
        _, _, offs = x.allgather(dist.hits_from_scratch(scs[0]), g.num_records)   # sizes the buffers, record counts
        assert offs == [0]
        works = []
        for i in range(5):                                   # rotation as in bench.py: two exchanges in flight
            sc = scs[i % 3]
            eng.ScanGenomeHits(g, cp, sc)
            while len(works) >= 2:
                x.finish(works.pop(0))
            works.append(x.start_scratch(sc, g.num_records))
        for w in works:
            x.finish(w)
        hits, ranges, offs = x.gathered()
        local = dist.hits_from_scratch(scs[4 % 3])
        assert ranges == [(0, len(local))] and offs == [0] and len(local) > 0
        key = lambda a: np.sort(a, order=["record", "pattern", "pos"])
        assert np.array_equal(key(np.unique(hits)), key(local))       # device order, maybe duplicated -> same set
        host = hip.engine.SimulationScratch(cp, host_only=True)
        lens = [g.record_len(r) for r in range(g.num_records)]
        flags = [g.record_flags(r) for r in range(g.num_records)]
        prods = eng.JoinHits(cp, host, hits, lens, flags)
        assert [p.sig() for p in prods] == want_products
        # more hits than the exchange capacity: the rank still enters the collective (first `cap` records, true count
        # in the header), finish() sees the overflow in the gathered header, regrows and redoes the exchange
        x2 = dist.HitExchanger(device=dev, cap_hits=max(1, len(local) // 2))
        x2.set_record_counts([g.num_records])
        eng.ScanGenomeHits(g, cp, scs[0])
        x2.finish(x2.start_scratch(scs[0], g.num_records))
        h2, r2, _ = x2.gathered()
        assert x2.redone == 1 and x2.cap >= len(local) and r2 == [(0, len(local))]
        assert np.array_equal(key(np.unique(h2)), key(local))
        x2.finish(x2.start_scratch(scs[0], g.num_records))            # now it fits: device form, nothing redone
        assert x2.redone == 1 and np.array_equal(key(np.unique(x2.gathered()[0])), key(local))
        assert x.native and x2.native                                 # RCCL inside the library (csrc/exchange.cpp) did all of the above
        # the same through torch.distributed (IPCR_EXCHANGE_TORCH=1: what a host without the native exchange runs), and
        # the overflow redo when the host's list is much shorter than the device's (the seed index files a window under
        # several keys; here: simulated by a list cut to ten records): the capacity must grow to the DEVICE count every rank
        # read in the headers, or every later exchange would overflow and be redone again
        monkeypatch.setenv("IPCR_EXCHANGE_TORCH", "1")
        x3 = dist.HitExchanger(device=dev, cap_hits=max(1, len(local) // 2))
        assert not x3.native
        x3.set_record_counts([g.num_records])
        x3.agree_on_device_path(scs[0])
        eng.ScanGenomeHits(g, cp, scs[0])
        monkeypatch.setattr(dist, "hits_from_scratch", lambda sc, _f=dist.hits_from_scratch: _f(sc)[:10])
        x3.finish(x3.start_scratch(scs[0], g.num_records))
        assert x3.redone == 1 and x3.cap >= len(local)
        monkeypatch.undo()
        monkeypatch.setenv("IPCR_EXCHANGE_SELFTEST", "1")
        x3.finish(x3.start_scratch(scs[0], g.num_records))
        assert x3.redone == 1 and np.array_equal(key(np.unique(x3.gathered()[0])), key(local))
        for xx in (x, x2, x3):
            xx.close()
    finally:
        if started_here and tdist.is_initialized():
            tdist.destroy_process_group()
    g.close()


def _raw_hit_records(hip, sc):
    """the scratch's host hit list (sorted, unique) as a HIT_DTYPE array"""
    from ipcr_amd import dist
    return dist.hits_from_scratch(sc)


def _fake_exchange(hip, world, rank, cap, rec_counts, same_records=False):
    import ctypes as C
    L = hip.lib.lib()
    uid = C.create_string_buffer(128)
    hip.lib.check(L.ipcr_exchange_unique_id(uid))
    x = C.c_void_p()
    hip.lib.check(L.ipcr_exchange_create(uid.raw, world, rank, 0, cap, int(same_records), C.byref(x)))
    hip.lib.check(L.ipcr_exchange_set_record_counts(x, (C.c_uint32 * world)(*rec_counts)))
    return x


def _exchange_end(hip, x, ticket, world):
    import ctypes as C
    import numpy as np
    from ipcr_amd.dist import HIT_DTYPE
    L = hip.lib.lib()
    hits, n = C.POINTER(hip.lib.Hit)(), C.c_int64(0)
    starts, offs = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
    hip.lib.check(L.ipcr_exchange_end(x, ticket, C.byref(hits), C.byref(n), C.byref(starts), C.byref(offs)))
    arr = np.zeros(0, dtype=HIT_DTYPE)
    if n.value:
        raw = (C.c_uint8 * (n.value * 32)).from_address(C.addressof(hits.contents))
        arr = np.frombuffer(raw, dtype=HIT_DTYPE, count=n.value).copy()
    return arr, [int(starts[r]) for r in range(world + 1)], [int(offs[r]) for r in range(world + 1)]


def _exchange_once(hip, x, sc, world=2):
    import ctypes as C
    t = C.c_int32(-1)
    hip.lib.check(hip.lib.lib().ipcr_exchange_begin(x, sc._h, C.byref(t)))
    return _exchange_end(hip, x, t.value, world)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_exchange_multi_rank_logic_over_a_fake_transport(hip, monkeypatch, world):
    """ipcr_exchange_begin / _end with world > 1 on one GPU: IPCR_TEST_EXCHANGE_FAKE replaces ncclAllGather by copies of this
    rank's block into every rank's place of the receive buffer, rank r's header rewritten to (hits >> r) records -- uneven
    counts, and for small scans zero-hit ranks.  Everything around the collective is the code a multi-GPU job runs, on
    device memory: shape agreement through the staging buffer, the strided read-back of the headers, the per-rank prefix
    copies, lock-step overflow + redo (two slots in flight), record rebasing, same_records.  (The collective itself runs
    with one rank in test_device_hit_exchange_one_rank_rccl; the unpack arithmetic alone on the CPU: test_exchange_unpack.py.)"""
    import ctypes as C
    import numpy as np
    from ipcr_amd import workloads
    monkeypatch.setenv("IPCR_TEST_EXCHANGE_FAKE", "1")
    L = hip.lib.lib()
    rng = random.Random(78)
    pairs = workloads.c2_pairs()
    g, _ = build_planted_genome(hip, rng, 3, 600_000, pairs[:1], 0x5eed4445, junk_every=2)
    eng = hip.engine.New(hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12))
    cp = eng.CompilePanel(pairs)
    scs = [eng.NewSimulationScratch(cp) for _ in range(2)]
    nrec = g.num_records
    want_products = [p.sig() for p in eng.ScanGenome(g, cp, scs[0])]
    eng.ScanGenomeHits(g, cp, scs[1])
    n = len(_raw_hit_records(hip, scs[0]))
    assert n >= 8
    # world 1: the device list as it is (append order) -- the reference for every rank's part below
    x1 = _fake_exchange(hip, 1, 0, 4 * n, [nrec])
    dev_list, st1, _ = _exchange_once(hip, x1, scs[0], world=1)
    assert st1 == [0, n] and {bytes(h) for h in dev_list} == {bytes(h) for h in _raw_hit_records(hip, scs[0])}
    L.ipcr_exchange_destroy(x1)
    counts = [n >> r for r in range(world)]
    rec_counts = [nrec + r for r in range(world)]           # (ranks of a real job may hold different numbers of records)
    for same in (False, True):
        for cap in (4 * n, max(1, n // 3)):                 # the second overflows on rank 0 (and 1): every rank redoes
            x = _fake_exchange(hip, world, world - 1, cap, rec_counts, same_records=same)
            t0, t1 = C.c_int32(-1), C.c_int32(-1)
            hip.lib.check(L.ipcr_exchange_begin(x, scs[0]._h, C.byref(t0)))      # two in flight, ended in the order begun
            hip.lib.check(L.ipcr_exchange_begin(x, scs[1]._h, C.byref(t1)))
            assert L.ipcr_exchange_begin(x, scs[0]._h, C.byref(C.c_int32())) != 0  # a third is refused
            for t in (t0.value, t1.value):
                hits, starts, offs = _exchange_end(hip, x, t, world)
                assert starts == [sum(counts[:r]) for r in range(world + 1)]
                assert offs == ([0] * (world + 1) if same else [sum(rec_counts[:r]) for r in range(world + 1)])
                if t == t0.value:
                    for r in range(world):
                        part = hits[starts[r]:starts[r + 1]].copy()
                        part["record"] -= np.uint32(offs[r])
                        assert part.tobytes() == dev_list[:counts[r]].tobytes(), (world, same, cap, r)
            assert L.ipcr_exchange_redone(x) == (2 if cap < n else 0)
            assert L.ipcr_exchange_capacity(x) >= min(cap, n)
            if not same:   # rank 0's part joins to the single-GPU products; the whole list to `world` shifted copies of them
                host = hip.engine.SimulationScratch(cp, host_only=True)
                lens, flags = [], []
                for r in range(world):
                    lens += [g.record_len(i) for i in range(nrec)] + [1000] * r
                    flags += [g.record_flags(i) for i in range(nrec)] + [0] * r
                hits0, starts0, _ = _exchange_once(hip, x, scs[0], world)
                prods = eng.JoinHits(cp, host, hits0[:starts0[1]], lens, flags)
                assert [p.sig() for p in prods] == want_products
                host.close()
            L.ipcr_exchange_destroy(x)
    assert L.ipcr_exchange_available(0) == 1 and L.ipcr_exchange_available(99) == 0
    g.close()


def test_many_tiny_records_in_one_block(hip):
    """hundreds of short records (empty, shorter than a primer, a few kb) share tile blocks: the in-kernel
    verifier starts from the block's first record and walks to the candidate's; every record vs the oracle"""
    rng = random.Random(515)
    E, P = hip.engine, hip.primer.Pair
    pairs = hip.primer.AddSelfPairs([P("p", "ACGTTGCATGCAAGCT", "GGCCTTAAGGCCATAT", 0, 0)])
    fwd, rc = pairs[0].Forward, O.revcomp(pairs[0].Reverse).decode()
    seqs = []
    for r in range(400):
        n = rng.choice([0, 5, 15, 16, 40, 300, 1500, 3000, 9000])
        s = [rng.choice("ACGT") for _ in range(n)]
        if n >= 300 and rng.random() < 0.7:
            a = rng.randrange(0, n - 200)
            s[a:a + 16] = fwd
            s[a + 150:a + 166] = rc
            if rng.random() < 0.3:
                s[a + 3] = O.different_base(s[a + 3])
        if n >= 40 and rng.random() < 0.2:
            q = rng.randrange(n)
            s[q] = "N"
        if n >= 16 and rng.random() < 0.2:       # a site flush with the record end
            s[n - 16:n] = rc
        seqs.append("".join(s).encode())
    g = E.Genome(sum(len(s) for s in seqs) + 8192 * (len(seqs) + 2), max_records=len(seqs) + 1)
    for r, s in enumerate(seqs):
        g.add_record("chr%d" % (r + 1), s)
    for cfg in (E.Config(MaxMM=1, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12),
                E.Config(MaxMM=2, TerminalWindow=0, MaxLen=0, HitCap=0, SeedLen=12, Circular=True)):
        _, _, sc, got = scan_and_compare(hip, cfg, pairs, g, seqs)
        assert sc.stats().kernel_kind == 1 and len(got) > 100
    g.close()


def test_jit_disk_cache(hip, tmp_path, monkeypatch):
    """IPCR_JIT_CACHE_DIR: the code object of a panel is written once and loaded by the next panel with the
    same source (another engine object, same primers and k)"""
    monkeypatch.setenv("IPCR_JIT_CACHE_DIR", str(tmp_path))
    E, P = hip.engine, hip.primer.Pair
    pairs = [P("disk", "ACGTTGCAAGGCTTAA", "TTGGCCAATTGGAACC", 0, 0)]
    seq = (b"TTTT" + b"ACGTTGCAAGGCTTAA" + b"ACGT" * 20 + O.revcomp(b"TTGGCCAATTGGAACC") + b"GGGG") * 3
    cfg = E.Config(MaxMM=1, TerminalWindow=2, MaxLen=500, HitCap=100, SeedLen=12)
    first = check(hip, cfg, seq, pairs)
    files = sorted(f.name for f in tmp_path.iterdir())
    assert len(files) >= 1 and all(f.endswith(".jit") for f in files)
    again = check(hip, cfg, seq, pairs)          # served from memory or disk, same answer
    assert [p.sig() for p in again] == [p.sig() for p in first] and sorted(f.name for f in tmp_path.iterdir()) == files
    # every file carries its own key (hiprtc version, arch, source) in front of the code object; a file whose stored
    # key differs (hash collision, another ROCm release) is ignored and rebuilt, not loaded
    raw = (tmp_path / files[0]).read_bytes()
    klen = int.from_bytes(raw[8:16], "little")
    assert raw[:8] == b"IPCRJIT1" and raw[16:23] == b"hiprtc " and b"ipcr_filter" in raw[16:16 + klen]
    assert raw[16 + klen:16 + klen + 4] == b"\x7fELF"
    (tmp_path / files[0]).write_bytes(raw[:16] + bytes([raw[16] ^ 1]) + raw[17:])   # stored key no longer matches
    pairs2 = [P("disk2", "ACGTTGCAAGGCTTAA", "TTGGCCAATTGGAACC", 0, 0)]             # same source, new panel object
    third = check(hip, cfg, seq, pairs2)
    assert [p.sig()[1:] for p in third] == [p.sig()[1:] for p in first]


def test_fallback_when_hand_over_does_not_arrive(tmp_path):
    """If the sweep's sequence word never shows up in pinned memory (simulated: it is written to device memory
    instead), the scan notices once the stream has drained, switches the process to the copy path and rescans;
    results are the same.  Runs in a child process: the switch is process-wide."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import ipcr_oracle as O
        from ipcr_amd import engine, primer
        seq = (b"TTTT" + b"ACGTTGCAAGGCTTAA" + b"ACGT" * 30 + O.revcomp(b"TTGGCCAATTGGAACC") + b"GGGG") * 5
        pairs = [primer.Pair("p", "ACGTTGCAAGGCTTAA", "TTGGCCAATTGGAACC", 0, 0)]
        cfg = engine.Config(MaxMM=1, TerminalWindow=2, MaxLen=500, HitCap=100, SeedLen=12)
        eng = engine.New(cfg); cp = eng.CompilePanel(pairs); sc = eng.NewSimulationScratch(cp)
        want = [w.sig() for w in O.simulate_batch(O.Config(max_mm=1, terminal_window=2, max_len=500, hit_cap=100, seed_len=12), seq,
                                                  [O.Pair("p", pairs[0].Forward, pairs[0].Reverse, 0, 0)])]
        for i in range(3):
            got = [p.sig() for p in eng.SimulateCompiledWithScratch("s", seq, cp, sc)]
            assert got == want and len(got) >= 5, (i, got, want)
        print("OK", sc.stats().kernel_kind)
    """ % (root, os.path.join(root, "oracle")))
    env = dict(os.environ, IPCR_TEST_BREAK_PUBLISH="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK 1" in r.stdout, (r.stdout, r.stderr)
    assert "using the copy path" in r.stderr


_HANDOVER_CHILD = """
    import sys, random
    sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
    import ipcr_oracle as O
    from ipcr_amd import engine, primer, workloads
    from test_host_logic import rand_seq, plant
    rng = random.Random(11)
    pair = workloads.bench_pair(0)
    pairs = primer.AddSelfPairs([pair])
    cfg = engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    ocfg = O.Config(max_mm=2, terminal_window=5, max_len=2000, hit_cap=10000, seed_len=12)
    opairs = [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs]
    seqs = []
    for r in range(4):
        s = rand_seq(rng, 300000, junk=False)
        for t in range(12):
            a = 1000 + t * 20000 + rng.randrange(500)
            plant(rng, s, pair.Forward, a, rng.choice([0, 1, 2]))
            plant(rng, s, O.revcomp(pair.Reverse).decode(), a + 160, 0)
        seqs.append("".join(s).encode())
    g = engine.Genome(sum(map(len, seqs)) + 65536, max_records=8)
    for r, s in enumerate(seqs):
        g.add_record("chr%%d" %% r, s)
    eng = engine.New(cfg); cp = eng.CompilePanel(pairs)
    op = O.Panel(ocfg, opairs)
    want = []
    for r, s in enumerate(seqs):
        want += [("chr%%d" %% r,) + w.sig() for w in op.scan(s)]
    scs = [eng.NewSimulationScratch(cp) for _ in range(3)]
    refetched = checked = diffs = 0
    for i in range(6):                                   # plain scans
        got = eng.ScanGenome(g, cp, scs[i %% 3])
        assert [(p.SequenceID,) + p.sig() for p in got] == want and len(want) >= 40, i
        st = scs[i %% 3].stats()
        assert st.kernel_kind == 1
        refetched += st.handover_refetched; checked += st.handover_checked; diffs += st.handover_check_diffs
    eng.ScanGenomeBegin(g, cp, scs[0])                    # chained sweeps, as bench.py runs them
    for i in range(9):
        if i + 1 < 9:
            scs[(i + 1) %% 3].chain_after(scs[i %% 3])
            eng.ScanGenomeBegin(g, cp, scs[(i + 1) %% 3])
        assert eng.ScanGenomeEndCount(g, cp, scs[i %% 3]) == len(want), i
        assert [(p.SequenceID,) + p.sig() for p in scs[i %% 3].products(g.ids)] == want
        st = scs[i %% 3].stats()
        refetched += st.handover_refetched; checked += st.handover_checked; diffs += st.handover_check_diffs
    print("OK refetched=%%d checked=%%d diffs=%%d hits=%%d" %% (refetched, checked, diffs, scs[0].stats().hits))
"""


def _run_handover_child(extra_env):
    import subprocess, sys, os, textwrap, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(_HANDOVER_CHILD % (root, os.path.join(root, "oracle"), os.path.join(root, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra_env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK refetched=" in r.stdout, (r.stdout, r.stderr[-3000:])
    m = re.search(r"OK refetched=(\d+) checked=(\d+) diffs=(\d+) hits=(\d+)", r.stdout)
    return tuple(int(x) for x in m.groups()), r.stderr


def test_handover_torn_record_is_not_accepted():
    """A hit record reaches pinned host memory as two 16-byte stores that nothing orders.  Simulated tear: the first
    half of record 2 carries a stale tag while its second half (the one the round-1 host looked at) is current.
    The host must not take it: it waits 2 ms, fetches the prefix from device memory instead (the straggler branch of
    scan_collect), and every result still equals the oracle's.  Child process: the knob is read once per process."""
    (refetched, _, _, hits), _ = _run_handover_child({"IPCR_TEST_WITHHOLD_TAG": "3"})
    assert refetched == 15 and hits >= 40          # every one of the 15 scans had to take the device copy


def test_handover_matches_device_memory():
    """IPCR_DEBUG_PUBLISH_CHECK=1: after every scan the records the host took from pinned memory are compared with
    what the kernel left in device memory -- zero differences, nothing refetched, plain and chained sweeps."""
    (refetched, checked, diffs, hits), err = _run_handover_child({"IPCR_DEBUG_PUBLISH_CHECK": "1"})
    assert checked == 15 and diffs == 0 and refetched == 0 and hits >= 40, err[-2000:]


# ---- one host process, several devices through the C ABI ------------------------------------------

def test_bind_thread_to_device_moves_only_the_calling_thread(hip):
    """ipcr_bind_thread_to_device: the calling thread ends up on a subset of the CPUs it was allowed on (the device's
    local_cpulist), nobody else moves, and 0 is an honest answer on a host with nothing to choose."""
    import os
    import threading
    from ipcr_amd import _lib as L
    main_before = os.sched_getaffinity(0)
    seen = {}

    def worker():
        before = os.sched_getaffinity(0)
        r = L.lib().ipcr_bind_thread_to_device(0)
        seen["r"], seen["before"], seen["after"] = r, before, os.sched_getaffinity(0)

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["r"] in (0, 1)
    assert seen["after"] <= seen["before"] and len(seen["after"]) > 0
    if seen["r"] == 1:
        assert len(seen["after"]) < len(seen["before"])
    else:
        assert seen["after"] == seen["before"]
    assert os.sched_getaffinity(0) == main_before
    assert L.lib().ipcr_bind_thread_to_device(10_000) == 0      # no such device: nothing happens



def test_one_process_drives_several_device_slots(hip, monkeypatch):
    """The reference's unit of parallelism is a pool of workers calling ForEachCompiledProduct on independent chunks
    (internal/pipeline/pipeline.go:60-125); over several GPUs that is worker i -> device i mod N with no collective.
    On the one-GPU test box IPCR_DEVICE_SLOTS adds device slots (each with panel tables and kernels of its own, all on
    the physical GPU): one compiled panel, scratches on three slots, worker threads that never select a device -- every
    entry point selects its object's device itself -- and results equal to the oracle on every slot."""
    import threading
    monkeypatch.setenv("IPCR_DEVICE_SLOTS", "3")
    E, P, L = hip.engine, hip.primer.Pair, hip.lib
    assert L.lib().ipcr_device_count() >= 3
    cfg = E.Config(MaxMM=2, TerminalWindow=3, MaxLen=400, HitCap=10000, SeedLen=12)
    pairs = hip.primer.AddSelfPairs([P("p", "ACGTTGCATGCAAGCT", "GGCCTTAAGGCCATAT")])
    eng = E.New(cfg)
    cp = eng.CompilePanel(pairs)
    rng = random.Random(5)
    jobs = []
    for j in range(12):
        n = rng.choice([4000, 60000, 300000])
        s = rand_case(rng, n, with_junk=(j % 3 == 0))
        for _ in range(3):
            a = rng.randrange(0, n - 400)
            plant(rng, s, pairs[0].Forward, a, rng.choice([0, 1, 2]))
            plant(rng, s, O.revcomp(pairs[0].Reverse).decode(), a + rng.randint(40, 300), 0)
        jobs.append("".join(s).encode())
    want = [[w.sig() for w in O.simulate_batch(ocfg(cfg), s, opairs(pairs))] for s in jobs]
    slots = [0, 1, 2, 1]
    scs = [eng.NewSimulationScratch(cp, device=d) for d in slots]
    assert [sc.device for sc in scs] == slots
    got, errors = [None] * len(jobs), []

    def worker(w):     # a fresh thread: no ipcr_set_device, no hipSetDevice
        try:
            for j in range(w, len(jobs), len(scs)):
                got[j] = [p.sig() for p in eng.SimulateCompiledWithScratch("job%d" % j, jobs[j], cp, scs[w])]
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(len(scs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and got == want and sum(map(len, want)) >= 12
    assert cp.device_slots == 3                       # three sets of tables, built at the first scan on each slot
    # resident genomes belong to a device too: a scratch of another device is refused, not silently misused
    g2 = E.Genome(1 << 20, 4, device=2)
    assert g2.device == 2
    g2.add_record("r0", jobs[1])
    assert [p.sig() for p in eng.ScanGenome(g2, cp, scs[2])] == want[1]
    with pytest.raises(L.IpcrError) as ei:
        eng.ScanGenome(g2, cp, scs[0])
    assert ei.value.status == L.ERR_INVALID
    with pytest.raises(L.IpcrError):
        eng.NewSimulationScratch(cp, device=7)
    g2.close()
    for sc in scs:
        sc.close()
    cp.close()


def test_device_slots_large_panel_and_native_pool(hip, monkeypatch, tmp_path):
    """the seed-index kernel's image and entry table are per-slot state as well; and the native worker pool
    (csrc/chunk_workers.cpp --devices) spreads its workers over the listed devices"""
    import json
    import os
    import subprocess
    from ipcr_amd import workloads
    monkeypatch.setenv("IPCR_DEVICE_SLOTS", "2")
    rng = random.Random(31)
    pairs = workloads.c4_pairs(160)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = hip.engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    seq = rand_case(rng, 400_000, with_junk=True)
    for t in range(20):
        p = pairs[t * 7 % 160]
        a = 2000 + t * 19000
        plant(rng, seq, p.Forward, a, t % 3)
        plant(rng, seq, O.revcomp(p.Reverse).decode(), a + 150, 0)
    seq = "".join(seq).encode()
    want = [w.sig() for w in O.simulate_batch(ocfg(cfg), seq, opairs(pairs))]
    for d in (1, 0):
        sc = eng.NewSimulationScratch(cp, device=d)
        assert [p.sig() for p in eng.SimulateCompiledWithScratch("s", seq, cp, sc)] == want and len(want) >= 15
        assert sc.stats().kernel_kind == 3
        sc.close()
    assert cp.device_slots == 2
    cp.close()
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ipcr_amd", "chunk_workers")
    r = subprocess.run([exe, "--devices", "0,1", "6000000", "1000000", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, IPCR_DEVICE_SLOTS="2"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["devices"] == [0, 1] and out["panel_device_slots"] == 2 and out["products_per_pass"] >= out["planted"] >= 5
    assert subprocess.run([exe, "1000000", "1500"], capture_output=True).returncode == 2      # chunk <= overlap: usage error
    assert subprocess.run([exe, "1000000", "500000", "0"], capture_output=True).returncode == 2  # no workers


# ---- the whole range of --mismatches the ABI accepts (IPCR_MAX_MM = 16) -----------------------------

@pytest.mark.parametrize("k", [5, 8, 16])
@pytest.mark.parametrize("kind", ["specialised", "specialised_long", "table_driven"])
def test_large_k_on_every_kernel(hip, k, kind):
    """k = 5, 8, 16 (the seed index stops at k = 3; beyond it panels go to the specialised filter in groups, or to
    filter_generic_quad_kernel<K1 = k + 1>, two patterns per walk up to k = 7, one beyond): primers of up to 32 nt (the specialised filter verifies its own survivors), primers of
    40-60 nt (filtered by their 20 positions next to the protected end, every survivor through the stand-alone verifier)
    and the table-driven kernel, on random sequence with junk bytes and sites planted with up to k substitutions outside
    the 3' window; HitCap 10000 and 0; vs the oracle (core/primer/match.go:67-84, core/engine/ac.go:186-213)"""
    rng = random.Random(1000 * k + len(kind))
    E, P = hip.engine, hip.primer.Pair
    lo, hi = (40, 60) if kind == "specialised_long" else ((26, 32) if kind == "specialised" else (28, 50))
    def mk():
        L = rng.randint(lo, hi)
        s = [rng.choice("ACGT") for _ in range(L)]
        if rng.random() < 0.3:
            s[rng.randrange(3, L - 4)] = rng.choice("RYSWKMN")
        return "".join(s)
    pairs = [P("p%d" % i, mk(), mk(), 0, 0) for i in range(2)]
    n = 40_000
    seq = rand_case(rng, n, with_junk=True)
    for t in range(12):
        p = pairs[t % 2]
        a = 500 + t * 3000
        def put(site, pos, nmut, protect_right):
            site = list(site)
            L = len(site)
            free = list(range(0, L - 4)) if protect_right else list(range(4, L))
            for j in rng.sample(free, min(nmut, len(free))):
                site[j] = rng.choice([c for c in "ACGT" if c != site[j]])
            seq[pos:pos + L] = site
        fwd = [c if c in "ACGT" else "A" for c in p.Forward]
        put(fwd, a, rng.choice([0, 1, k // 2, k]), True)
        rc = [c if c in "ACGT" else "A" for c in O.revcomp(p.Reverse).decode()]
        put(rc, a + 200, rng.choice([0, k // 2, k]), False)
    text = "".join(seq).encode()
    for hit_cap in (10000, 0):
        cfg = E.Config(MaxMM=k, TerminalWindow=3, MaxLen=400, HitCap=hit_cap, SeedLen=12)
        eng = E.New(cfg)
        cp = eng.CompilePanel(pairs)
        if kind == "table_driven":
            cp.set_specialize(False)
        sc = eng.NewSimulationScratch(cp)
        got = eng.SimulateCompiledWithScratch("seq", text, cp, sc)
        st = sc.stats()
        assert st.kernel_kind == (2 if kind == "table_driven" else 1)
        if kind == "specialised_long":
            assert st.verify_ms > 0          # primers beyond 32 nt: every survivor goes through the stand-alone verifier
        want = O.simulate_batch(ocfg(cfg), text, opairs(pairs))
        assert [g.sig() for g in got] == [w.sig() for w in want]
        assert len(want) >= 6, len(want)
        sc.close()
        cp.close()


@pytest.mark.parametrize("k,tw,cap", [(0, 3, 50), (1, 3, 7), (2, 0, 1000)])
def test_hit_cap_bounds_device_memory(hip, monkeypatch, k, tw, cap):
    """A low-complexity primer on low-complexity sequence: millions of raw matches of which HitCap per orientation and
    record can matter (core/primer/match.go:86-88, core/engine/hit_collect.go:80-82).  Beyond a soft limit of raw hit
    records (lowered here: IPCR_TEST_HCAP_SOFT) the scan is not regrown but repeated in position order over ranges of
    blocks, the host keeping what the cap can use (host.cpp: scan_segmented): the products equal the oracle's, the
    5'-window / cap-before-filter quirks of the rc orientations included, and the device hit buffer stays small."""
    monkeypatch.setenv("IPCR_TEST_HCAP_SOFT", "30000")
    E, P = hip.engine, hip.primer.Pair
    rng = random.Random(9 + k)
    recs = []
    for r in range(3):
        s = bytearray(b"A" * 400_000)
        for _ in range(40):                     # a few other bases and junk bytes: mismatches, reset bytes, record-specific caps
            i = rng.randrange(len(s))
            s[i] = rng.choice(b"CGTNn")
        if r == 1:
            s[1000:1012] = b"TTTTTTTTTTTT"
        recs.append(bytes(s))
    pairs = [P("polyA", "AAAAAAAAAAAA", "TTTTTTTTTTTT"), P("mixed", "AAAAAACAAAAA", "TTTTTTTTTTGT")]
    cfg = E.Config(MaxMM=k, TerminalWindow=tw, MaxLen=60, HitCap=cap, SeedLen=12)
    eng = E.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    g = E.Genome(sum(map(len, recs)) + (1 << 16), 4)
    for r, s in enumerate(recs):
        g.add_record("r%d" % r, s)
    got = eng.ScanGenome(g, cp, sc)
    st = sc.stats()
    assert st.segmented == 1
    import ctypes as C
    L = hip.lib.lib()
    ptr, nh, hcap = C.c_void_p(), C.c_uint64(), C.c_uint64()
    # the device buffer holds the LAST range only: a caller that would read it (an all-gather straight out of it) is told so
    assert L.ipcr_scratch_device_hits(sc._h, C.byref(ptr), C.byref(nh), C.byref(hcap)) == hip.lib.ERR_UNSUPPORTED
    assert hcap.value <= (1 << 20)              # the initial buffer: never regrown towards the raw match count
    # ... and the library's own exchange sends the host list instead (fake transport: rank r reports hits >> r)
    monkeypatch.setenv("IPCR_TEST_EXCHANGE_FAKE", "1")
    kept = {bytes(h) for h in _raw_hit_records(hip, sc)}
    for xcap in (1 << 16, 16):                  # the second: smaller than the list -> lock-step redo from the host list again
        x = _fake_exchange(hip, world=2, rank=0, cap=xcap, rec_counts=[3, 3])
        hits, starts, offs = _exchange_once(hip, x, sc)
        assert starts[1] == len(kept) and starts[2] - starts[1] == len(kept) >> 1 and offs[:2] == [0, 3]
        assert {bytes(h) for h in hits[:starts[1]]} == kept
        second = hits[starts[1]:starts[2]].copy()
        second["record"] -= 3
        assert second.tobytes() == hits[:len(second)].tobytes()
        assert L.ipcr_exchange_redone(x) == (1 if len(kept) > xcap else 0)   # (cap 7: the whole list fits 16 slots)
        L.ipcr_exchange_destroy(x)
    monkeypatch.delenv("IPCR_TEST_EXCHANGE_FAKE")
    want = []
    for r, s in enumerate(recs):
        for w in O.simulate_batch(ocfg(cfg), s, opairs(pairs)):
            want.append((r,) + w.sig())
    assert [(p.Record,) + p.sig() for p in got] == want and len(want) >= 10
    # a scan in rolling windows cannot be cut out of that list -- the device kept per RECORD what HitCap can use, every window has
    # a cap of its own: the library says so and the caller streams the windows through ipcr_scan_chunk (ipcr_amd/cli.py does)
    with pytest.raises(hip.lib.IpcrError) as e:
        eng.ScanGenomeChunked(g, cp, sc, 100_000, 100)
    assert e.value.status == hip.lib.ERR_UNSUPPORTED
    # the chunk path goes the same way (one record may fit the buffer as it is)
    got1 = eng.SimulateCompiledWithScratch("r1", recs[1], cp, sc)
    assert [p.sig() for p in got1] == [w.sig() for w in O.simulate_batch(ocfg(cfg), recs[1], opairs(pairs))]
    g.close(); sc.close(); cp.close()


def test_disk_cache_of_code_objects_is_bounded(hip, tmp_path, monkeypatch):
    """The code objects persist on disk by default; the directory must not grow without limit: after a new file is written
    the least recently used ones go until the files fit IPCR_JIT_CACHE_MAX_MB (here 1 MB: two C2-sized kernels do not fit),
    a hit touches its file, and every file's stored key names the HIP runtime version (a code object of another ROCm build
    is never loaded)."""
    import glob, os, time
    from ipcr_amd import workloads
    d = tmp_path / "cache"
    d.mkdir()
    monkeypatch.setenv("IPCR_JIT_CACHE_DIR", str(d))
    monkeypatch.setenv("IPCR_JIT_CACHE_MAX_MB", "1")
    monkeypatch.setenv("IPCR_JIT_NO_MEMCACHE", "1")
    seq = O.bench_dna(200_000, 5)
    cfg = hip.engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
    sizes = []
    first = None
    for i in range(4):
        pairs = hip.primer.AddSelfPairs([workloads.bench_pair(40 + i)])
        hip.engine.New(cfg).SimulateBatch("s", seq, pairs)
        files = sorted(glob.glob(str(d / "ipcr_*.jit")), key=os.path.getmtime)
        assert files, "no code object was written"
        total = sum(os.path.getsize(f) for f in files)
        sizes.append((len(files), total))
        if first is None:
            first = files[0]
            blob = open(first, "rb").read(400)
            assert blob[:8] == b"IPCRJIT1" and b"hiprtc " in blob and b" runtime " in blob and b"gfx950" in blob
        assert total <= (1 << 20) or len(files) == 1, sizes
        time.sleep(0.05)
    assert not os.path.exists(first)                      # the oldest went
    assert not glob.glob(str(d / "*.tmp*"))
    # a hit touches its file: scanning the newest panel again makes it the most recently used
    newest = sorted(glob.glob(str(d / "ipcr_*.jit")), key=os.path.getmtime)[-1]
    before = os.path.getmtime(newest)
    time.sleep(0.05)
    hip.engine.New(cfg).SimulateBatch("s", seq, hip.primer.AddSelfPairs([workloads.bench_pair(43)]))
    assert os.path.getmtime(newest) > before


def test_small_panel_scans_before_its_kernels_are_built(tmp_path):
    """IPCR_JIT_ASYNC (the default outside the tests): hiprtc builds a small panel's kernels on a thread of its own; a scan
    that comes before they are ready takes the table-driven kernel, later ones the specialised one; same products."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import ipcr_oracle as O
        from ipcr_amd import engine, primer, workloads
        pairs = workloads.c2_pairs()
        seq = bytearray(O.bench_dna(300000, 77))
        for a in (1000, 90000, 250000):
            seq[a:a + 20] = pairs[0].Forward.encode()
            seq[a + 160:a + 180] = O.revcomp(pairs[0].Reverse)
        seq = bytes(seq)
        cfg = engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)
        want = [w.sig() for w in O.simulate_batch(O.Config(max_mm=2, terminal_window=5, max_len=2000, hit_cap=10000, seed_len=12), seq,
                                                  [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs])]
        eng = engine.New(cfg); cp = eng.CompilePanel(pairs); sc = eng.NewSimulationScratch(cp)
        t0 = time.perf_counter()
        got = [p.sig() for p in eng.SimulateCompiledWithScratch("s", seq, cp, sc)]
        t1 = time.perf_counter()
        k1 = sc.stats().kernel_kind
        assert got == want and len(got) >= 3, (got, want)
        cp.wait_ready()
        t2 = time.perf_counter()
        got = [p.sig() for p in eng.SimulateCompiledWithScratch("s", seq, cp, sc)]
        assert got == want and sc.stats().kernel_kind == 1
        print("OK", k1, round(t1 - t0, 3), round(t2 - t0, 3))
    """ % (root, os.path.join(root, "oracle")))
    env = dict(os.environ, IPCR_JIT_ASYNC="1", IPCR_JIT_CACHE_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout, r.stderr[-3000:])
    _, k1, first_s, built_s = r.stdout.split()
    assert k1 == "2" and float(first_s) < float(built_s)   # cold cache: the first scan did not wait for hiprtc
