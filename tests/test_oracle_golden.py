"""Pins the CPU oracle against every literal known-answer the reference's own tests hold
for the hot path.  Each test names the reference test it restates."""
from collections import Counter

import pytest

import ipcr_oracle as O
from ipcr_oracle import Config, Pair


def multiset(products):
    return Counter(p.sig() for p in products)


# ---- core/primer ---------------------------------------------------------------------------

def test_base_match_table():  # core/primer/iupac_test.go:6-25
    for g, p, want in [("A", "A", True), ("G", "R", True), ("C", "R", False), ("T", "N", True),
                       ("G", "N", True), ("A", "B", False), ("C", "B", True), ("T", "X", False)]:
        assert O.base_match(g, p) is want


def test_iupac_mask_snapshot():  # core/primer/iupac_snapshot_test.go:5-22
    assert [O.iupac_mask(c) for c in "ACGT"] == [1, 2, 4, 8]
    assert O.iupac_mask("U") == O.iupac_mask("T") and O.iupac_mask("u") == O.iupac_mask("t")
    assert (O.iupac_mask("R"), O.iupac_mask("Y"), O.iupac_mask("N")) == (1 | 4, 2 | 8, 15)
    assert O.iupac_mask("r") == O.iupac_mask("R") and O.iupac_mask("n") == O.iupac_mask("N")


def test_genome_non_acgt_is_hard_mismatch():  # core/primer/iupac.go:62-67
    for g in "NRYacgtn-":
        assert not O.base_match(g, "N")


def test_find_matches_table():  # core/primer/match_test.go:6-76
    seq = "ACGTACGTACGT"
    for primer, mm, tw, count, first in [("ACG", 0, 0, 3, 0), ("AGG", 1, 0, 3, 0), ("AGG", 0, 0, 0, -1),
                                         ("ACA", 1, 1, 0, -1), ("ACG", 1, 0, 3, 0), ("ACN", 0, 0, 3, 0)]:
        hits = O.find_matches(seq, primer, mm, 0, tw)
        assert len(hits) == count
        if count:
            assert hits[0].pos == first


def test_mismatch_count():  # core/primer/mismatch_test.go:6-33
    for w, p, want in [("ACGT", "ACGT", 0), ("ACGT", "NNNN", 0), ("ACGT", "RRRR", 2), ("ACGT", "TTTT", 3)]:
        assert O.mismatch_count(w, p) == want
    with pytest.raises(ValueError):
        O.mismatch_count("AAA", "AA")


def test_revcomp():  # core/primer/rc_test.go:9-52, rc_snapshot_test.go:7-16
    assert O.revcomp("AGTC") == b"GACT"
    assert O.revcomp("RYSWKMBDHVN") == b"NBDHVKMWSRY"
    assert O.revcomp("RYSWKMBDHVNACGT") == b"ACGTNBDHVKMWSRY"
    assert O.revcomp("") == b""
    for bad in ("ACGX", "acgt"):
        with pytest.raises(ValueError):
            O.revcomp(bad)


# ---- core/engine ---------------------------------------------------------------------------

def test_simulate_minimal():  # core/engine/engine_test.go:11-35 + SURVEY appendix A.1
    got = O.simulate_batch(Config(), "ACGTACGTACGT", [Pair("test", "ACG", "ACG")])
    assert (got[0].start, got[0].end, got[0].length) == (0, 12, 12)
    coords = [(p.type, p.start, p.end, p.length) for p in got]
    six = [(0, 12, 12), (0, 8, 8), (0, 4, 4), (4, 12, 8), (4, 8, 4), (8, 12, 4)]
    assert coords == [("forward",) + c for c in six] + [("revcomp",) + c for c in six]


def test_length_filtering():  # engine_test.go:38-78
    got = O.simulate_batch(Config(), "ACGTACGTACGT", [Pair("t", "ACG", "ACG", 10, 12)])
    assert [(p.type, p.start, p.end) for p in got] == [("forward", 0, 12), ("revcomp", 0, 12)]
    assert O.simulate_batch(Config(), "ACGTACGTACGT", [Pair("t2", "ACG", "ACG", 5, 7)]) == []


def test_revcomp_product():  # engine_test.go:81-101 + appendix A.2
    got = O.simulate_batch(Config(), "TTTACGACGTAAA", [Pair("rev", "ACG", "TTT")])
    assert [(p.type, p.start, p.end, p.length) for p in got] == [
        ("forward", 3, 13, 10), ("forward", 6, 13, 7), ("revcomp", 0, 10, 10)]


def test_circular_amplicon():  # engine_test.go:104-129 + appendix A.3
    pair = [Pair("p1", "AG", "TC")]
    assert O.simulate_batch(Config(circular=False), "TGACAAG", pair) == []
    got = O.simulate_batch(Config(circular=True), "TGACAAG", pair)
    assert len(got) == 1
    assert (got[0].type, got[0].start, got[0].end, got[0].length) == ("forward", 5, 3, 5)


def test_seeded_mismatch_protected_window():  # engine_test.go:131-148 + appendix A.4
    got = O.simulate_batch(Config(max_mm=1, terminal_window=3, seed_len=12, min_len=10),
                           "CAGTACAAAAAAGGTACC", [Pair("seed-mm", "AAGTAC", "GGTACC")])
    assert len(got) == 1
    assert (got[0].start, got[0].end, got[0].length, got[0].fwd_mm, got[0].fwd_idx, got[0].rev_idx) == \
        (0, 18, 18, 1, (0,), ())


def test_seeded_mismatch_no_tw():  # engine_test.go:150-168
    cfg = Config(max_mm=1, terminal_window=0, seed_len=12, min_len=10)
    pair = [Pair("seed-mm-no-tw", "AAGTAC", "GGTACC")]
    panel = O.Panel(cfg, pair)
    assert panel.num_seed_patterns > 0 and panel.have(0, "A")
    assert len(panel.scan("CAGTACAAAAAAGGTACC")) == 1


def test_appendix_a6_full_product_list():  # approx_seed_oracle_test.go:114-122,187 (hand-derived list)
    got = O.simulate_batch(Config(max_mm=1, terminal_window=3, min_len=1, max_len=100, seed_len=12),
                           "TTTTCGTACAAAAGGTACCTTT", [Pair("x", "ACGTAC", "GGTACC")])
    assert multiset(got) == Counter([
        ("x", 3, 19, 16, "forward", 1, 0, (0,), ()),
        ("x", 12, 19, 7, "forward", 1, 0, (1,), ()),
        ("x", 13, 20, 7, "revcomp", 0, 1, (), (1,)),
    ])


def test_ac_stream():  # core/engine/ac_test.go:9-31
    assert O.ac_scan(["ACG", "CG"], "TTACGNCGacgACNG") == [(4, 0), (4, 1), (7, 1), (10, 0), (10, 1)]


def test_non_acgt_ranges():  # core/engine/non_acgt_halo_test.go:9-15
    assert O.non_acgt_ranges("aaNaaRRtt") == [(2, 3), (5, 7)]


def test_halo_starts_merge():  # non_acgt_halo_test.go:17-30
    assert O.halo_starts(20, 6, [(5, 6), (7, 8)]) == list(range(8))


def test_halo_used_for_seeded_orientation():  # non_acgt_halo_test.go:32-52
    pairs = [Pair("reference_n_inside_seed", "ACGTAC", "GGTACC")]
    cfg = Config(max_mm=1, terminal_window=0, min_len=1, max_len=100, seed_len=6)
    panel = O.Panel(cfg, pairs)
    assert panel.have(0, "A")
    seq = "TTTACNTACAAAAGGTACCTTT"
    got = panel.scan(seq)
    assert multiset(got) == multiset(O.simulate_bruteforce(cfg, seq, pairs))
    assert len(got) >= 1


def test_seed_dedup_counts():  # core/engine/seed_dedup_test.go:8-58
    pairs = [Pair("p1", "AAAACCCC", "GGGGTTTT"), Pair("p2", "AAAACCCC", "GGGGTTTT")]
    panel = O.Panel(Config(max_mm=0, terminal_window=0, seed_len=4), pairs)
    pats = dict(panel.seed_patterns())
    assert len(pats) == 4
    for s in ("CCCC", "TTTT", "GGGG", "AAAA"):
        assert pats[s] == 2
    for i in range(2):
        for w in "ABab":
            assert panel.have(i, w)
    assert O.build_seed_patterns_count(pairs, 4, 0, 1) == 52
    p1 = O.Panel(Config(max_mm=1, terminal_window=0, seed_len=4), pairs)
    assert any(n > 1 for _, n in p1.seed_patterns())


def test_seed_len_over_32_falls_back():  # seed_dedup_test.go:87-104
    pairs = [Pair("long_seed", "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT",
                  "TGCATGCATGCATGCATGCATGCATGCATGCATGCATGCA")]
    panel = O.Panel(Config(seed_len=33), pairs)
    assert panel.num_seed_patterns == 0
    assert not any(panel.have(0, w) for w in "ABab")


def test_variant_explosion_falls_back_per_orientation():  # performance_gate_test.go:25-49
    pairs = [Pair("variant_cap", "NNNNNNNNNNNN", "ACGTACGTACGT"), Pair("ordinary", "ACGTACGTACGT", "TGCATGCATGCA")]
    panel = O.Panel(Config(max_mm=2, terminal_window=0, min_len=1, max_len=100, seed_len=12), pairs)
    assert not panel.have(0, "A") and not panel.have(0, "a")
    assert panel.have(0, "B") and panel.have(0, "b")
    assert all(panel.have(1, w) for w in "ABab")


def test_regression_gate_all_seeded():  # performance_gate_test.go:8-23
    seq, pairs = O.make_bench_fixture(16, 20000, True, False)
    panel = O.Panel(Config(max_mm=1, terminal_window=0, min_len=100, max_len=240, seed_len=12), pairs)
    assert panel.num_seed_patterns > 0 and panel.num_nodes > 1
    assert all(panel.have(i, w) for i in range(16) for w in "ABab")


def test_collector_semantics_through_engine():  # core/engine/hit_collect_test.go:27-41 (cap keeps first start)
    cfg = Config(max_mm=0, hit_cap=1, seed_len=2)
    panel = O.Panel(cfg, [Pair("x", "AA", "AA")])
    ms = panel.scan_matches("AAAAAA", 0, "A")
    assert [m.pos for m in ms] == [0]


def test_scratch_reuse_does_not_leak():  # hit_collect_test.go:98-112
    panel = O.Panel(Config(max_mm=0, min_len=6, max_len=60, seed_len=4), [Pair("x", "ACGTAC", "GGTACC")])
    assert len(panel.scan("TTTACGTACAAAAGGTACCTTT")) > 0
    assert panel.scan("TTTACGTACAAAACCCCCCCTTT") == []


# ---- differential: production path vs brute-force oracle -----------------------------------

ORACLE_CASES = [  # core/engine/approx_seed_oracle_test.go:95-176
    ("TTTACGTACAAAAGGTACCTTT", [Pair("forward_exact", "ACGTAC", "GGTACC")]),
    ("TTTGGTACCAAAAGTACGTTTT", [Pair("revcomp_exact", "ACGTAC", "GGTACC")]),
    ("TTTTCGTACAAAAGGTACCTTT", [Pair("forward_mismatch_5prime", "ACGTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTT", [Pair("primer_ry", "ACRTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTT", [Pair("primer_internal_n", "ACNTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTT", [Pair("primer_3prime_n", "ACGTAN", "GGTACC")]),
    ("TTTNCGTACAAAAGGTACCTTT", [Pair("reference_n", "ACGTAC", "GGTACC")]),
    ("TTTacgtacAAAAGGTACCTTT", [Pair("lowercase_reference", "ACGTAC", "GGTACC")]),
    ("TTTACGTACAAAAGGTACCTTTGGGGGGGGGG", [Pair("panel_hit", "ACGTAC", "GGTACC"), Pair("panel_decoy", "TTAACC", "CCAATT")]),
]
ORACLE_CONFIGS = [Config(max_mm=k, terminal_window=tw, min_len=1, max_len=100, seed_len=12)
                  for k in (0, 1, 2) for tw in (0, 1, 3)]  # :178-191


@pytest.mark.parametrize("case", range(len(ORACLE_CASES)))
@pytest.mark.parametrize("ci", range(len(ORACLE_CONFIGS)))
def test_batch_matches_bruteforce(case, ci):  # approx_seed_oracle_test.go:88-203
    seq, pairs = ORACLE_CASES[case]
    cfg = ORACLE_CONFIGS[ci]
    assert multiset(O.simulate_batch(cfg, seq, pairs)) == multiset(O.simulate_bruteforce(cfg, seq, pairs))


def test_length_boundaries():  # approx_seed_oracle_test.go:205-226 + appendix A.5
    seq, pairs = "TTTACGTACAAAAGGTACCTTT", [Pair("length", "ACGTAC", "GGTACC")]
    for mn, mx, n in [(16, 16, 1), (17, 100, 0), (1, 15, 0)]:
        cfg = Config(min_len=mn, max_len=mx, seed_len=12)
        fast = O.simulate_batch(cfg, seq, pairs)
        assert multiset(fast) == multiset(O.simulate_bruteforce(cfg, seq, pairs))
        assert len(fast) == n
        if n:
            assert (fast[0].type, fast[0].start, fast[0].end, fast[0].length) == ("forward", 3, 19, 16)


def test_circular_vs_bruteforce():  # approx_seed_oracle_test.go:228-240
    for circ in (False, True):
        cfg = Config(circular=circ)
        pairs = [Pair("circular", "AG", "TC")]
        assert multiset(O.simulate_batch(cfg, "TGACAAG", pairs)) == multiset(O.simulate_bruteforce(cfg, "TGACAAG", pairs))


def test_self_pairs_vs_bruteforce():  # approx_seed_oracle_test.go:242-259
    seq, pairs = "TTTACGTACAAAAGTACGTTTT", [Pair("self+self", "ACGTAC", "ACGTAC")]
    for k, tw in [(0, 0), (1, 3), (2, 0)]:
        cfg = Config(max_mm=k, terminal_window=tw, min_len=1, max_len=100, seed_len=12)
        assert multiset(O.simulate_batch(cfg, seq, pairs)) == multiset(O.simulate_bruteforce(cfg, seq, pairs))


def test_benchmark_workload_matches_bruteforce():  # performance_gate_test.go:51-59
    seq, pairs = O.make_bench_fixture(12, 20000, True, True)
    cfg = Config(max_mm=2, terminal_window=0, min_len=100, max_len=240, seed_len=12)
    got = O.simulate_batch(cfg, seq, pairs)
    assert multiset(got) == multiset(O.simulate_bruteforce(cfg, seq, pairs))
    # every planted amplicon (180 bp, mutated idx 10 + reference N at idx 11) is recovered
    planted = {(p.experiment_id, p.start) for p in got if p.type == "forward" and p.length == 180}
    assert planted == {("bench_%03d" % i, 128 + 256 * i) for i in range(12)}
    for p in got:
        if p.length == 180 and p.type == "forward":
            assert p.fwd_idx == (10, 11) and p.fwd_mm == 2 and p.rev_mm == 0


def test_join_stream_fixture():  # core/engine/join_stream_test.go:23-46 (same inputs; oracle joins both ways)
    seq = "TTTACGTACAAAAGGTACCTTTGGGACGTATAAAAGGTACCAAA"
    pairs = [Pair("join_stream", "ACGTAC", "GGTACC", 6, 80)]
    cfg = Config(max_mm=1, min_len=6, max_len=80)
    assert multiset(O.simulate_batch(cfg, seq, pairs)) == multiset(O.simulate_bruteforce(cfg, seq, pairs))


# ---- fixtures ------------------------------------------------------------------------------

def test_lcg_fixture_generators():  # performance_benchmark_test.go:67-106
    x = 0x5eed1234
    want = []
    for _ in range(64):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        want.append("ACGT"[(x >> 30) & 3])
    assert O.bench_dna(64, 0x5eed1234).decode() == "".join(want)
    for idx in (0, 1, 7, 2047):
        x = (0x9e3779b9 ^ (idx * 0x45d9f3b)) & 0xFFFFFFFF
        buf = []
        for i in range(20):
            x = (x * 1103515245 + 12345 + i * 97) & 0xFFFFFFFF
            buf.append("ACGT"[(x >> 29) & 3])
        buf[0], buf[1], buf[2], buf[19] = "ACGT"[idx & 3], "ACGT"[(idx + 1) & 3], "ACGT"[(idx + 2) & 3], "ACGT"[(idx + 3) & 3]
        assert O.bench_primer(idx, 20) == "".join(buf)
    assert [O.different_base(b) for b in "ACGTN"] == list("CGTAA")


# ---- core/oligo, core/probe ----------------------------------------------------------------

def test_best_hit():  # core/oligo/oligo_test.go:5-21, core/probe/annotate_test.go:5-19
    h = O.best_hit("ACGTACGTACGT", "GTAC", 0)
    assert (h.found, h.pos, h.mm, h.strand, h.site) == (True, 2, 0, "+", "GTAC")
    assert O.best_hit("ACGTACGTACGT", "GTGC", 1).found
    h = O.best_hit("AAAGACCC", "GAY", 0)
    assert (h.found, h.strand, h.pos, h.site) == (True, "+", 3, "GAC")
    assert not O.best_hit("ACGT", "  ", 0).found


def test_sort_matches_by_pos_and_bounds():  # core/engine/match_search_test.go:8-31
    pos, order, lo, hi = O.sort_and_bounds([8, 2, 5, 5], 5)
    assert pos == [2, 5, 5, 8]
    assert (lo, hi) == (1, 3)
    assert order == [1, 2, 3, 0]          # sort.SliceStable (engine.go:80-83): the two 5s keep their input order
    assert O.sort_and_bounds([1, 2, 3], 0)[2:] == (0, 0) and O.sort_and_bounds([1, 2, 3], 9)[2:] == (3, 3)
    assert O.sort_and_bounds([], 4) == ([], [], 0, 0)


def test_validate_primer():  # core/primer/validate_test.go:5-25
    assert O.validate_primer(" acgtry swkmbdhvn ") == "ACGTRYSWKMBDHVN"
    with pytest.raises(ValueError):
        O.validate_primer("ACGX")
    with pytest.raises(ValueError):
        O.validate_primer("ACGU")          # :21 -- U is rejected at the input boundary, although iupacMask knows it
    with pytest.raises(ValueError):
        O.validate_primer(" '\" ")
    from ipcr_amd import primer            # the host mirror the CLI uses follows the same rule
    assert primer.Validate(" acgtry swkmbdhvn ") == "ACGTRYSWKMBDHVN"
    for bad in ("ACGX", "ACGU", ""):
        with pytest.raises(ValueError):
            primer.Validate(bad)
