"""Batched nested PCR (ipcr_nested_windows / ipcr_nested_products) against the reference's visitor semantics
(internal/visitors/nested.go:17-66): per outer amplicon, SimulateBatch of the inner pairs and the best product
by (total mismatches, -length, start, end, pair ID)."""
import random

import pytest

import ipcr_oracle as O

pytestmark = pytest.mark.gpu


def best_inner(cfg, amp: bytes, inner_pairs):
    hits = O.simulate_batch(cfg, amp, inner_pairs)
    if not hits:
        return None
    hits = sorted(hits, key=lambda h: (h.fwd_mm + h.rev_mm, -h.length, h.start, h.end, h.experiment_id))
    h = hits[0]
    return (h.experiment_id, h.start, h.end, h.length, h.type, h.fwd_mm, h.rev_mm)


def test_nested_visit_chooses_best_after_sorting():  # internal/visitors/nested_test.go:9-39
    from ipcr_amd import engine, nested, primer
    seq = b"AAAACACACACGGGACACACTTTACCCC"
    g = engine.Genome(1 << 16, 4)
    g.add_record("amp", seq)
    eng = engine.New(engine.Config(MaxMM=0, TerminalWindow=0))
    cp = eng.CompilePanel([primer.Pair("short-late", "TTT", "GGG"), primer.Pair("long-early", "AAA", "CCC")])
    sc = eng.NewSimulationScratch(cp)
    got = nested.NestedWindows(g, [(0, 0, len(seq))], cp, sc)[0]
    assert got.InnerFound and got.InnerPairID == "long-early" and got.InnerStart == 0 and got.InnerLength == 14
    g.close()


def test_nested_products_vs_oracle():
    from ipcr_amd import engine, nested, primer
    rng = random.Random(404)
    outer = primer.Pair("outer", "ACGTTGCATGCAAGCTTAGC", "GGCCTTAAGGCCATATCGTA")
    inner = [primer.Pair("in1", "TTGACCGATTAC", "CCGGTTAACGGA"), primer.Pair("in2", "GATTACAGGTCA", "ACGGATTCAGGC"),
             primer.Pair("in3", "TTGACCGATTMC", "CCGGTTAACGGR")]
    rc_o = O.revcomp(outer.Reverse).decode()
    recs = []
    for r in range(3):
        s = list(O.bench_dna(300_000, 900 + r).decode())
        for t in range(15):
            a = 2000 + t * 19000
            ln = rng.choice([300, 600, 1200])
            s[a:a + 20] = outer.Forward
            s[a + ln - 20:a + ln] = rc_o
            for ip in rng.sample(inner[:2], rng.choice([0, 1, 2])):       # inner amplicons inside the outer one
                b = a + 30 + rng.randrange(40)
                iln = rng.choice([80, 150, 200])
                f = list(ip.Forward)
                if rng.random() < 0.4:
                    f[2] = O.different_base(f[2])
                s[b:b + 12] = f
                s[b + iln - 12:b + iln] = O.revcomp(ip.Reverse).decode()
            if t == 7:
                s[a + 100:a + 104] = "NNNN"
        recs.append("".join(s).encode())
    g = engine.Genome(1_200_000, 4)
    for r, b in enumerate(recs):
        g.add_record("chr%d" % r, b)
    ocfg = engine.Config(MaxMM=1, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = engine.New(ocfg)
    cpo = eng.CompilePanel([outer])
    sco = eng.NewSimulationScratch(cpo)
    prods = eng.ScanGenome(g, cpo, sco)
    assert len(prods) >= 40
    for icfg in (engine.Config(MaxMM=1, TerminalWindow=3, MaxLen=2000, HitCap=10000, SeedLen=12),
                 engine.Config(MaxMM=0, TerminalWindow=0, MinLen=100, MaxLen=0, HitCap=0, SeedLen=12)):
        ieng = engine.New(icfg)
        cpi = ieng.CompilePanel(inner)
        sci = ieng.NewSimulationScratch(cpi)
        got = nested.NestedProducts(sco, prods, g, cpi, sci)
        oc = O.Config(max_mm=icfg.MaxMM, terminal_window=icfg.TerminalWindow, min_len=icfg.MinLen, max_len=icfg.MaxLen,
                      hit_cap=icfg.HitCap, seed_len=icfg.SeedLen)
        opairs = [O.Pair(p.ID, p.Forward, p.Reverse, 0, 0) for p in inner]
        found = 0
        for p, n in zip(prods, got):
            amp = recs[p.Record][p.Start:p.End]
            want = best_inner(oc, amp, opairs)
            have = (n.InnerPairID, n.InnerStart, n.InnerEnd, n.InnerLength, n.InnerType, n.InnerFwdMM, n.InnerRevMM) if n.InnerFound else None
            assert have == want, (p, have, want)
            found += n.InnerFound
        assert found >= 10
        assert len(nested.NestedProducts(sco, prods, g, cpi, sci, require_inner=True)) == found
    g.close()


def test_nested_large_batch_matches_per_window_calls():
    """4000 windows (overlapping, of many lengths, some empty) in one batch == the same windows one call each for a
    sample: the batched pack places every amplicon where its record table says"""
    from ipcr_amd import engine, nested, primer
    rng = random.Random(808)
    seq = list(O.bench_dna(400_000, 4242).decode())
    inner = [primer.Pair("i1", "TTGACCGATTAC", "CCGGTTAACGGA"), primer.Pair("i2", "GATTACAGGTCA", "ACGGATTCAGGC")]
    for t in range(300):
        ip = inner[t % 2]
        b = 500 + t * 1300
        seq[b:b + 12] = ip.Forward
        seq[b + 90:b + 102] = O.revcomp(ip.Reverse).decode()
    g = engine.Genome(500_000, 2)
    g.add_record("r", "".join(seq).encode())
    eng = engine.New(engine.Config(MaxMM=1, TerminalWindow=2, MaxLen=1000, HitCap=100, SeedLen=12))
    cp = eng.CompilePanel(inner)
    sc = eng.NewSimulationScratch(cp)
    windows = []
    for _ in range(4000):
        a = rng.randrange(0, 399_000)
        windows.append((0, a, a + rng.choice([0, 7, 60, 130, 400, 900])))
    got = nested.NestedWindows(g, windows, cp, sc)
    assert sum(n.InnerFound for n in got) >= 300
    for i in rng.sample(range(4000), 60):
        one = nested.NestedWindows(g, [windows[i]], cp, sc)[0]
        assert (one.InnerFound, one.InnerPairID, one.InnerStart, one.InnerEnd, one.InnerFwdMM, one.InnerRevMM) == \
            (got[i].InnerFound, got[i].InnerPairID, got[i].InnerStart, got[i].InnerEnd, got[i].InnerFwdMM, got[i].InnerRevMM)
    g.close()
