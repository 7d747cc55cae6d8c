"""CPU tests of the host side of the product: the C ABI surface, the panel compiler's
seeded/unseeded decisions, and the hit -> match-list -> join logic (ordering, HitCap, 5' window
filter, circular wrap) checked against the oracle with hits synthesised on the CPU.
No device is touched: scans need a GPU and must fail loudly without one."""
import os
import random
import re

import numpy as np
import pytest

import ipcr_oracle as O
from ipcr_amd import _lib, dist, engine, primer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ipcr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ipcr_[a-z0-9_]+)\s*\(", hdr)) - {"ipcr_emit_fn"}
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    L = _lib.lib()  # resolves all of them or raises
    assert L.ipcr_version().startswith(b"ipcr-hip")
    import ctypes as C
    assert C.sizeof(_lib.Hit) == 32 and C.sizeof(_lib.Product) == 24 + 7 * 4 + 32 + 4  # padded to 8


def test_core_primer_helpers():  # core/primer/iupac_test.go, rc_test.go through the ABI
    for g, p, want in [("A", "A", True), ("G", "R", True), ("C", "R", False), ("T", "N", True),
                       ("A", "B", False), ("C", "B", True), ("T", "X", False), ("N", "N", False), ("a", "A", False)]:
        assert primer.BaseMatch(g, p) is want
    assert primer.RevComp("RYSWKMBDHVNACGT") == b"ACGTNBDHVKMWSRY"
    assert primer.RevComp("") == b""
    for bad in ("ACGX", "acgt"):
        with pytest.raises(_lib.IpcrError):
            primer.RevComp(bad)
    assert primer.Validate(" ac'g\"t\n") == "ACGT"
    with pytest.raises(ValueError):
        primer.Validate("ACGU")


def test_no_device_fails_loudly():
    if _lib.lib().ipcr_device_count() > 0:
        pytest.skip("a GPU is present")
    cp = engine.New(engine.Config()).CompilePanel([primer.Pair("x", "ACGT", "ACGT")])
    with pytest.raises(_lib.IpcrError) as e:
        engine.SimulationScratch(cp)
    assert e.value.status == _lib.ERR_DEVICE and "no CPU fallback" in e.value.message
    with pytest.raises(_lib.IpcrError):
        engine.New(engine.Config()).SimulateBatch("s", b"ACGTACGT", [primer.Pair("x", "ACGT", "ACGT")])
    host = engine.SimulationScratch(cp, host_only=True)
    import ctypes as C
    st = _lib.lib().ipcr_scan_chunk(cp._h, host._h, b"ACGT", 4, None, None)
    assert st == _lib.ERR_DEVICE


def rand_primer(rng, lo=4, hi=30, amb=True):
    L = rng.randint(lo, hi)
    s = [rng.choice("ACGT") for _ in range(L)]
    if amb:
        for _ in range(rng.choice([0, 0, 1, 2, 6])):
            s[rng.randrange(L)] = rng.choice("RYSWKMBDHVN")
    return "".join(s)


def test_have_matches_reference_seed_rules():  # core/engine/seed.go:152-367 via compiled.go:123-133
    rng = random.Random(42)
    for _ in range(300):
        pairs = [primer.Pair("p%d" % i, rand_primer(rng), rand_primer(rng)) for i in range(rng.randint(1, 3))]
        if rng.random() < 0.1:
            pairs.append(primer.Pair("n", "N" * rng.randint(8, 14), rand_primer(rng)))
        k, tw, sl = rng.choice([0, 1, 2, 3]), rng.choice([0, 1, 3, 5, 40]), rng.choice([0, 12, 6, 4, 33, -1])
        cp = engine.New(engine.Config(MaxMM=k, TerminalWindow=tw, SeedLen=sl)).CompilePanel(pairs)
        op = O.Panel(O.Config(max_mm=k, terminal_window=tw, seed_len=sl),
                     [O.Pair(p.ID, p.Forward, p.Reverse) for p in pairs])
        for i in range(len(pairs)):
            for w in "ABab":
                assert cp.have(i, w) == op.have(i, w), (pairs[i], w, k, tw, sl)
        cp.close()
        op.close()
    # literal expectations of core/engine/performance_gate_test.go:25-49
    cp = engine.New(engine.Config(MaxMM=2, SeedLen=12)).CompilePanel(
        [primer.Pair("variant_cap", "NNNNNNNNNNNN", "ACGTACGTACGT"), primer.Pair("ordinary", "ACGTACGTACGT", "TGCATGCATGCA")])
    assert [cp.have(0, w) for w in "ABab"] == [False, True, False, True]
    assert all(cp.have(1, w) for w in "ABab")


def synth_hits(cp, seq, k, record, mode):
    """What the device would report for one record: every distinct scanned pattern's verified
    matches (core/primer/match.go:30-90 semantics via the oracle), as ipcr_hit records."""
    used = {cp.slot_pattern(i, w, mode) for i in range(len(cp.Pairs)) for w in "ABab"}
    used &= set(cp.scanned_patterns(mode))    # a pattern shard (ipcr_panel_set_shard) scans a slice of them
    rows = []
    for gid in sorted(used):
        pat, left, tw_dev, soff, slen = cp.pattern_info(gid)
        if not pat:
            continue
        if left:
            ms = [m for m in O.find_matches(seq, pat, k, 0, 0) if all(j >= tw_dev for j in m.idx)]
        else:
            ms = O.find_matches(seq, pat, k, 0, tw_dev)
        for m in ms:
            flag = 0
            if slen:
                span = seq[m.pos + soff:m.pos + soff + slen]
                flag = int(any(ch not in b"ACGTacgt" for ch in span))
            m0 = sum(1 << j for j in m.idx if j < 64)
            m1 = sum(1 << (j - 64) for j in m.idx if j >= 64)
            rows.append((m.pos, record, gid | (flag << 31), m0, m1))
    rng = random.Random(len(rows))
    rng.shuffle(rows)  # the device appends in no particular order
    return np.array(rows, dtype=dist.HIT_DTYPE) if rows else np.zeros(0, dtype=dist.HIT_DTYPE)


def rand_seq(rng, n, junk):
    s = [rng.choice("ACGT") for _ in range(n)]
    if junk:
        for _ in range(rng.randint(1, 5)):
            p, run, ch = rng.randrange(n), rng.randint(1, 9), rng.choice("NNRacgtn")
            for i in range(p, min(n, p + run)):
                s[i] = ch
    return s


def plant(rng, seq, pat, pos, nmut):
    conc = [rng.choice([b for b in "ACGT" if O.base_match(b, ch)]) for ch in pat]
    for _ in range(nmut):
        j = rng.randrange(len(conc))
        conc[j] = O.different_base(conc[j])
    seq[pos:pos + len(conc)] = conc


@pytest.mark.parametrize("seed", range(8))
def test_join_matches_reference_production_path(seed):
    """ipcr_join_hits == Engine.ForEachCompiledProduct (core/engine/compiled.go:162-267), product
    for product and in the same order, for random panels, junk bytes, caps and circular mode."""
    rng = random.Random(900 + seed)
    for _ in range(25):
        n = rng.choice([80, 400, 3000])
        nrec = rng.randint(1, 3)
        pairs = [primer.Pair("p%d" % i, rand_primer(rng, 4 if n < 1000 else 8, 24), rand_primer(rng, 4 if n < 1000 else 8, 24),
                             rng.choice([0, 0, 15]), rng.choice([0, 0, 300])) for i in range(rng.randint(1, 3))]
        if rng.random() < 0.5:
            pairs = primer.AddSelfPairs(pairs)
        k = rng.choice([0, 1, 2, 3] if n < 1000 else [0, 1, 2])
        cfg = engine.Config(MaxMM=k, TerminalWindow=rng.choice([0, 1, 3, 5]), MinLen=rng.choice([0, 8]),
                            MaxLen=rng.choice([0, 150, 2000]), HitCap=rng.choice([0, 0, 1, 3, 10000]),
                            SeedLen=rng.choice([0, 12, 5, -1]), Circular=rng.random() < 0.35)
        seqs = []
        for r in range(nrec):
            s = rand_seq(rng, n, junk=rng.random() < 0.6)
            for p in pairs:
                for _ in range(rng.randint(0, 3)):
                    a = rng.randrange(0, n - 60)
                    ln = rng.randint(len(p.Forward) + len(p.Reverse), 58)
                    plant(rng, s, p.Forward, a, rng.choice([0, 0, 1, 2]))
                    rc = O.revcomp(p.Reverse).decode()
                    plant(rng, s, rc, a + ln - len(rc), rng.choice([0, 0, 1]))
            seqs.append("".join(s).encode())
        eng = engine.New(cfg)
        cp = eng.CompilePanel(pairs)
        reset = [any(ch not in b"ACGTacgt" for ch in s) for s in seqs]
        mode = 1 if any(reset) else 0          # what ipcr_scan_genome would pick for this genome
        flags = [(1 if reset[r] else 0) | (2 if mode else 0) for r in range(nrec)]
        hits = np.concatenate([synth_hits(cp, seqs[r], k, r, mode) for r in range(nrec)])
        sc = engine.SimulationScratch(cp, host_only=True)
        got = eng.JoinHits(cp, sc, hits, [len(s) for s in seqs], flags, ["rec%d" % r for r in range(nrec)])
        op = O.Panel(O.Config(max_mm=cfg.MaxMM, terminal_window=cfg.TerminalWindow, min_len=cfg.MinLen,
                              max_len=cfg.MaxLen, hit_cap=cfg.HitCap, seed_len=cfg.SeedLen, circular=cfg.Circular),
                     [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs])
        want = []
        for r, s in enumerate(seqs):
            want += [("rec%d" % r,) + w.sig() for w in op.scan(s)]
        assert [(g.SequenceID,) + g.sig() for g in got] == want, (cfg, pairs)
        sc.close()
        cp.close()
        op.close()


def test_join_known_answers():  # SURVEY appendix A.1-A.4 through the join alone
    eng = engine.New(engine.Config())
    cp = eng.CompilePanel([primer.Pair("test", "ACG", "ACG")])
    seq = b"ACGTACGTACGT"
    sc = engine.SimulationScratch(cp, host_only=True)
    got = eng.JoinHits(cp, sc, synth_hits(cp, seq, 0, 0, 0), [len(seq)], [0])
    six = [(0, 12, 12), (0, 8, 8), (0, 4, 4), (4, 12, 8), (4, 8, 4), (8, 12, 4)]
    assert [(p.Type, p.Start, p.End, p.Length) for p in got] == [("forward",) + c for c in six] + [("revcomp",) + c for c in six]
    eng = engine.New(engine.Config(Circular=True))
    cp = eng.CompilePanel([primer.Pair("p1", "AG", "TC")])
    sc = engine.SimulationScratch(cp, host_only=True)
    got = eng.JoinHits(cp, sc, synth_hits(cp, b"TGACAAG", 0, 0, 0), [7], [0])
    assert [(p.Type, p.Start, p.End, p.Length) for p in got] == [("forward", 5, 3, 5)]


def test_panel_rejects_bad_input():
    E, P = engine, primer.Pair
    for cfg, pair, status in [(E.Config(MaxMM=-1), P("x", "ACGT", "ACGT"), _lib.ERR_INVALID),
                              (E.Config(MaxMM=17), P("x", "ACGT", "ACGT"), _lib.ERR_UNSUPPORTED),
                              (E.Config(), P("x", "ACGX", "ACGT"), _lib.ERR_PRIMER),
                              (E.Config(), P("x", "acgt", "ACGT"), _lib.ERR_PRIMER),
                              (E.Config(), P("x", "A" * 129, "ACGT"), _lib.ERR_UNSUPPORTED)]:
        with pytest.raises(_lib.IpcrError) as e:
            E.New(cfg).CompilePanel([pair])
        assert e.value.status == status


def test_filter_source_is_generated_for_the_headline_panels():
    from ipcr_amd import workloads
    cp = engine.New(engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)).CompilePanel(workloads.c2_pairs())
    assert cp.num_patterns == 4  # 3 pairs / 12 orientation slots / 4 distinct patterns
    src = cp.filter_source(0)
    assert "ipcr_filter" in src and src.count("// pattern 0:") > 0
    mid = engine.New(engine.Config(MaxMM=2, TerminalWindow=3)).CompilePanel(workloads.c4_pairs(24))
    assert "ipcr_filter" in mid.filter_source(0)  # 96 patterns: cut into 8 groups, one kernel each
    big = engine.New(engine.Config(MaxMM=2, TerminalWindow=3)).CompilePanel(workloads.c4_pairs(64))
    assert big.filter_source(0) == ""  # 256 patterns: too many kernels to compile, table-driven filter
    long_p = "ACGT" * 9  # 36 nt: filtered on the 20 positions next to the protected end; survivors go to the stand-alone verifier
    lsrc = engine.New(engine.Config(MaxMM=1, TerminalWindow=3)).CompilePanel([primer.Pair("l", long_p, long_p)]).filter_source(0)
    assert "ipcr_filter" in lsrc and "#define LIST_CAP 0u" in lsrc and "len 20, 3 protected" in lsrc
    # right-protected pattern: its window starts 16 rows before the filtered part (the push's constant says so: pattern id | 16 << 16)
    assert "u32 info = %du;" % (0 | (16 << 16)) in lsrc and "if (q == 1u) info = 1u;" in lsrc and "wp -= (u64)off" in lsrc
    # the form for small launches (mode 4 = set 0, a block shared by four waves): the rolled loop between a wave's own bounds, every
    # window test behind "it >= seg_e0", the head quads loaded by the wave itself, four waves per block in the ticket arithmetic
    seg = cp.filter_source(4)
    assert "const u64 nwv = nblocks * 4ull;" in seg and "for (u32 it = seg_f0; it < seg_e1; ++it)" in seg
    assert "const u32 seg_e0 = sg * 8u / 4u, seg_e1 = (sg + 1u) * 8u / 4u" in seg       # 37 quads in iterations of five: 8 iterations
    assert seg.count("it >= seg_e0") == 20 and "if (it == 0u) { st[" not in seg            # one guard per row of the 20-row loop body
    assert "if (seg_e1 * 5u + 2u > 32u) {" in seg and "st[0][lane] = own[0];" in seg
    assert "const u64 nwv = nblocks * 1ull;" in src and "seg_e0" not in src
    too_long = "ACGT" * 33           # 132 nt: beyond IPCR_MAX_PRIMER_LEN
    with pytest.raises(_lib.IpcrError):
        engine.New(engine.Config(MaxMM=1)).CompilePanel([primer.Pair("l", too_long, too_long)])


def test_generated_filter_structure():
    """what the generator decides for the headline panel: the coarse k+1 block split with the exact count in the rare
    branch, pattern tables that match the panel, and no exact stage once a kernel holds many patterns"""
    import re
    from ipcr_amd import workloads
    cp = engine.New(engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12)).CompilePanel(workloads.c2_pairs())
    src = cp.filter_source(0)
    assert len(re.findall(r"// pattern \d+: len 20, 5 protected, 3 blocks", src)) >= 4   # B = k + 1
    assert "__builtin_expect(all != 0xFFFFFFFFu, 0)" in src and "const u32 e14 =" in src   # exact count over 15 positions
    m = re.search(r"PMASK\[NPAT \* 32u\] = \{([0-9,]+)\}", src)
    table = [int(v) for v in m.group(1).split(",")]
    assert len(table) == 4 * 32
    fwd = workloads.c2_pairs()[0].Forward
    code = {"A": 1, "C": 2, "G": 4, "T": 8}
    want = [code[b] | (16 if j >= 15 else 0) for j, b in enumerate(fwd)] + [0] * 12    # 3' window = last five positions
    assert any(table[q * 32:(q + 1) * 32] == want for q in range(4))
    many = engine.New(engine.Config(MaxMM=2, TerminalWindow=3)).CompilePanel(workloads.c4_pairs(8))
    big_src = many.filter_source(0)                                                       # 12 patterns per kernel: > 160 sites
    assert "ipcr_filter" in big_src and "const u32 e14 =" not in big_src


def test_workload_primers_match_reference_generator():  # performance_benchmark_test.go:78-93
    from ipcr_amd import workloads
    for idx in (0, 1, 2, 3, 77, 2047):
        assert workloads.bench_primer(idx) == O.bench_primer(idx, 20)
    assert len(workloads.c4_pairs(1024)) == 1024 + 2048


@pytest.mark.parametrize("ns,pipeline,exchange", [(3, True, True), (2, True, True), (3, False, True), (3, True, False), (1, False, True), (4, True, True)])
def test_pipelined_pass_schedule_invariants(ns, pipeline, exchange):
    """bench.pipelined_passes with fake scratches and a fake exchanger: no scratch is begun while its previous pass
    is open or while an exchange still reads its device hit buffer, at most two exchanges are in flight, everything
    is finished at the end -- the hazards of the multi-GPU loop that no single-GPU run can show"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    class Scr:
        def __init__(self, name):
            self.name, self.open, self.read_by = name, False, None
    scs = [Scr(i) for i in range(ns)]
    log, inflight, ended = [], set(), []

    def begin(cur):
        assert not cur.open, "scan begun while the previous one on this scratch is still open"
        assert cur.read_by is None, "scan begun while exchange %s still reads this scratch's hit buffer" % cur.read_by
        cur.open = True
        log.append(("begin", cur.name))

    def end(i, cur):
        assert cur.open
        cur.open = False
        ended.append(i)
        return i * 10

    def chain(cur, prev):
        assert cur is not prev and not cur.open

    def start(cur):
        assert not cur.open and cur.read_by is None
        h = ("x", len(log))
        cur.read_by = h
        inflight.add(h)
        assert len(inflight) <= 2, "more exchanges in flight than receive slots"
        log.append(("start", cur.name))
        return (h, cur)

    def finish(w):
        h, cur = w
        assert h in inflight
        inflight.discard(h)
        cur.read_by = None

    if ns == 1 and pipeline:
        return
    k = 17
    last = bench.pipelined_passes(k, scs, begin, end, chain=chain if pipeline else None,
                                  start_exchange=start if exchange else None, finish_exchange=finish if exchange else None,
                                  pipeline=pipeline)
    assert last == (k - 1) * 10 and ended == list(range(k)) and not inflight
    assert all(not s.open and s.read_by is None for s in scs)
    assert bench.pipelined_passes(0, scs, begin, end) is None


def test_bench_launch_decision(monkeypatch):
    """`python bench.py --gpus N` must become N ranks: without RANK in the environment the process only launches
    torch.distributed.run as a child (fake launcher here) and returns its exit code; a process that already is a rank
    runs in place and refuses a --gpus that contradicts WORLD_SIZE"""
    import os, subprocess, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.launch_plan(8, {"RANK": "3", "WORLD_SIZE": "8"}, ["--gpus", "8"]) is None
    assert bench.launch_plan(1, {}, []) is None
    assert bench.launch_plan(1, {"RANK": "0", "WORLD_SIZE": "2"}, []) is None     # torchrun without --gpus: env decides
    cmd = bench.launch_plan(8, {}, ["--gpus", "8", "--steps", "5"], python="py", script="/x/bench.py")
    assert cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {"RANK": "0", "WORLD_SIZE": "2"}, [])
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append(cmd) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(v, raising=False)
    torch_loaded = "torch" in sys.modules
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and len(calls) == 1
    assert calls[0][-4:] == ["--gpus", "4", "--steps", "3"] and calls[0][calls[0].index("--nproc-per-node") + 1] == "4"
    assert ("torch" in sys.modules) == torch_loaded, "the launcher process must not import torch"


def _counter_blocks(src):
    """(number of flags, statements) of every block counter in a generated filter: the lines between a pattern's
    last `const u32 eN = ...;` and its `f |= ...;`"""
    out, flags, stmts, in_pat = [], 0, [], False
    for line in src.splitlines():
        t = line.strip()
        if t.startswith("{ // pattern") or t == "{":
            flags, stmts, in_pat = 0, [], True
            continue
        if not in_pat:
            continue
        m = re.match(r"const u32 e(\d+) = ", t)
        if m:
            flags, stmts = int(m.group(1)) + 1, []
            continue
        if re.match(r"(const u32 x\d+|u32 u\d+|u\d+ \|?=) ", t) or re.match(r"const u32 x\d+ = .*, x\d+ = ", t):
            stmts.append(t)
            continue
        if t.startswith("f = ANDOR(") and flags:   # thermometer: the top level goes straight into the verdict, once per flag
            stmts.append(t)
            continue
        if (t.startswith("f |= ") or t.startswith("f0 = f") or re.match(r"f\d+(_\d)? = f;", t) or t == "}") and flags and (t.startswith("f |= ") or any(x.startswith("f = ANDOR(") for x in stmts)):
            out.append((flags, stmts + ([t] if t.startswith("f |= ") else [])))
            flags, stmts, in_pat = 0, [], False
    return out


@pytest.mark.parametrize("k,tw,primers", [
    (1, 3, ("ACGTTGCATGCAAGCTAGCT", "GGCCTTAAGGCCATATGGCA")),
    (2, 5, ("ACGTTGCATGCAAGCTAGCT", "GGCCTTAAGGCCATATGGCA")),
    (3, 3, ("AGAGTTTGATCMTGGCTCAG", "TACGGYTACCTTGTTAYGACTT")),
    (3, 0, ("ACGTTGCATGCAAGCTAGCTAGGATC", "GGCCTTAAGGCCATATGGCATTACGGA")),
    (4, 2, ("ACGTTGCATGCAAGCTAGCTAGGATCAA", "GGCCTTAAGGCCATATGGCATTACGGAT")),
    (5, 0, ("ACGTTGCATGCAAGCTAGCTAGGATCAATT", "GGCCTTAAGGCCATATGGCATTACGGATCC")),
    # the range the ABI accepts beyond what the index serves (IPCR_MAX_MM = 16): wide counters, long primers
    (6, 3, ("ACGTTGCATGCAAGCTAGCTAGGATCAATT", "GGCCTTAAGGCCATATGGCATTACGGATCC")),
    (8, 2, ("ACGTTGCATGCAAGCTAGCTAGGATCAATTGGCATGCAAT", "GGCCTTAAGGCCATATGGCATTACGGATCCAAGTCCGTAA")),
    (12, 0, ("ACGTTGCATGCAAGCTAGCTAGGATCAATTGGCATGCAATTTGACCA", "GGCCTTAAGGCCATATGGCATTACGGATCCAAGTCCGTAACCGATTG")),
    (16, 3, ("ACGTTGCATGCAAGCTAGCTAGGATCAATTGGCATGCAATTTGACCAGGT", "GGCCTTAAGGCCATATGGCATTACGGATCCAAGTCCGTAACCGATTGACC")),
])
def test_generated_block_counters_are_exact(k, tw, primers, monkeypatch):
    """the "more than k of the B block flags are set" circuits the generator emits (thermometer or carry-save adder
    tree, whichever is cheaper on v_bitop3) evaluated for EVERY combination of flags, main test and exact stage"""
    for force in ("0", "1", "2"):
        monkeypatch.setenv("IPCR_JIT_COUNTER", force)
        cp = engine.New(engine.Config(MaxMM=k, TerminalWindow=tw, MaxLen=2000)).CompilePanel(
            [primer.Pair("p", primers[0], primers[1], 0, 0)])
        src = cp.filter_source(0)
        cp.close()
        blocks = _counter_blocks(src)
        # (with k >= the unprotected positions of the 20-position filter window nothing is counted there: any number of
        # mismatches passes the filter and the exact verifier decides)
        assert blocks or k >= 16, "no counter found in the generated source"
        seen = set()
        for flags, stmts in blocks:
            key = (flags, tuple(stmts))
            if key in seen or flags > 20:   # (a filter window has 20 positions: 2^20 assignments at most)
                continue
            seen.add(key)
            n = 1 << flags
            full = (1 << n) - 1
            env = {"f": 0, "FULL": full, "XOR3": lambda a, b, c: a ^ b ^ c, "MAJ3": lambda a, b, c: (a & b) | (c & (a | b)),
                   "ANDOR": lambda a, b, c: (a & b) | c}
            assign = np.arange(n, dtype=np.uint32)      # every assignment of the flags, as a number
            as_int = lambda bits: int.from_bytes(np.packbits(bits, bitorder="little").tobytes(), "little")
            for i in range(flags):     # flag i as a truth-table column over all 2^flags assignments
                env["e%d" % i] = as_int(((assign >> i) & 1).astype(np.uint8))
            for st in stmts:
                st = st.rstrip(";").replace("const u32 ", "").replace("u32 ", "").replace("0u", "0")
                for part in re.split(r",\s*(?=x\d+ = )", st):
                    part = re.sub(r"~(\w+)", r"(FULL ^ \1)", part)
                    exec(part, {}, env)
            ones = np.zeros(n, dtype=np.uint8)
            for i in range(flags):
                ones += ((assign >> i) & 1).astype(np.uint8)
            want = as_int((ones > k).astype(np.uint8))
            assert env["f"] & full == want, (k, flags, force, stmts)


def _slot0_patterns(src):
    """the generated main loop's code for the windows that end at register slot 0: per pattern the leaves -- (plane letter,
    slot) -- of its protected OR and of every block flag, temporaries resolved"""
    lines = src.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("  for (u32 it = "))
    start = next(i for i in range(start, len(lines)) if "// window slot 0" in lines[i])
    out, cur, tmp = [], None, {}
    leaf = re.compile(r"\b([acgtn])(\d+)\b")
    ref = re.compile(r"\bo\d+_\d+\b")
    def leaves(expr):
        got = [(m.group(1), int(m.group(2))) for m in leaf.finditer(expr)]
        for r in ref.findall(expr):
            got += tmp[r]
        return got
    for l in lines[start:]:
        if "// window slot 1" in l:
            break
        t = l.strip()
        m = re.match(r"\{ // pattern (\d+): len (\d+), (\d+) protected, (\d+) blocks", t)
        if m:
            cur = {"q": int(m.group(1)), "len": int(m.group(2)), "nprot": int(m.group(3)), "nblocks": int(m.group(4)), "prot": [], "blocks": []}
            tmp = {}
            continue
        if cur is None:
            continue
        m = re.match(r"const u32 (o\d+_\d+) = (.*);", t)
        if m:
            tmp[m.group(1)] = leaves(m.group(2))
            continue
        m = re.match(r"u32 f = (.*);", t)
        if m:
            cur["prot"] = leaves(m.group(1))
            continue
        m = re.match(r"const u32 e\d+ = (.*);", t)
        if m:
            cur["blocks"].append(leaves(m.group(1)))
            continue
        if re.match(r"f\d+ = f;", t):
            out.append(cur)
            cur = None
    return out


@pytest.mark.parametrize("rebalance", ["1", "0"])
@pytest.mark.parametrize("k,tw,primers", [
    (2, 5, ("ACGTTGCATGGATCCTAACG", "TTGACCGTAGGCATTCAGGA")),
    (3, 3, ("AGAGTTTGATCMTGGCTCAG", "TACGGYTACCTTGTTAYGAC")),
    (1, 3, ("ACGTNGCATGCAAGCTAGCT", "GGCCTTRAGGCCATATGGYA")),
    (3, 0, ("ACBTTGCATGCAAGDTAGCT", "GGMCTTAAGGCCWTATGGCA")),
    (4, 2, ("ACGTTGCATSCAAGCTAG", "GGCCTKAAGGCCATATGGCA")),
    (2, 3, ("ACGTTGCATGCAAG", "GGMCTYAAGGCCRTAT")),
])
def test_filter_blocks_partition_the_window(k, tw, primers, rebalance, monkeypatch):
    """the specialised filter is sound only if every position of a pattern's window is tested exactly once -- in the
    protected OR or in exactly one block -- through exactly the mismatch planes its IUPAC code calls for (`x` = base is
    not X; a code of several bases is the AND of their planes, N the invalid plane).  Which positions share a block is
    the generator's choice (jit.cpp: make_plan moves positions between blocks while the instruction count falls,
    IPCR_JIT_REBALANCE): read it back from the generated source, for windows ending at slot 0 of the main loop"""
    monkeypatch.setenv("IPCR_JIT_REBALANCE", rebalance)
    cfg = engine.Config(MaxMM=k, TerminalWindow=tw, MaxLen=2000)
    cp = engine.New(cfg).CompilePanel(primer.AddSelfPairs([primer.Pair("p", primers[0], primers[1], 0, 0)]))
    src = cp.filter_source(0)
    assert src
    ids = cp.scanned_patterns(0)
    pats = _slot0_patterns(src)
    assert len(pats) == len(ids) >= 2
    infos = [cp.pattern_info(i) for i in ids]
    lmax = max(len(x[0]) for x in infos)
    assert lmax <= 20
    W = (lmax + 3) // 4 * 4
    sr = (0 - (lmax - 1)) % W
    for pat, (seq, left, tw_dev, _so, _sl) in zip(pats, infos):
        L = len(seq)
        assert pat["len"] == L
        want = {}
        for j, ch in enumerate(seq):
            allowed = [b for b in "ACGT" if O.base_match(b, ch)]
            slot = (sr + j) % W
            want[j] = {("n", slot)} if len(allowed) == 4 else {(b.lower(), slot) for b in allowed}
        groups = [pat["prot"]] + pat["blocks"]
        flat = [x for g in groups for x in g]
        assert len(flat) == len(set(flat)), (seq, "a plane is read twice")
        assert set(flat) == set().union(*want.values()), (seq, "the window's positions and planes")
        for j in range(L):   # a position's planes stay together: one group holds them all
            assert sum(1 for g in groups if want[j] <= set(g)) == 1, (seq, j)
        prot_pos = set(range(tw_dev)) if left else set(range(L - tw_dev, L))
        got_prot = {j for j in range(L) if want[j] <= set(pat["prot"])}
        if k > 0 and L - len(prot_pos) > k:
            assert got_prot == prot_pos, (seq, left, tw_dev, got_prot)
            assert len(pat["blocks"]) == pat["nblocks"] >= k + 1
    cp.close()


def test_pattern_shards_join_to_the_unsharded_result():
    """ipcr_panel_set_shard: the distinct patterns are dealt round-robin over `count` panel objects; their hit lists
    over the same records, concatenated in any order, join (full panel) to exactly the unsharded products -- the
    orientations of a pair are independent until the per-pair join (core/engine/compiled.go:192-207,260-265)"""
    rng = random.Random(4711)
    pairs = primer.AddSelfPairsUnique([primer.Pair("p%d" % i, "".join(rng.choice("ACGT") for _ in range(18)),
                                                   "".join(rng.choice("ACGT") for _ in range(20)), 0, 0) for i in range(5)])
    cfg = engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=400, HitCap=50, SeedLen=12)
    seqs = []
    for r in range(3):
        s = rand_seq(rng, 6000, junk=(r == 1))
        for i in range(5):
            a = 200 + i * 1000
            plant(rng, s, pairs[i].Forward, a, rng.choice([0, 1, 2]))
            plant(rng, s, O.revcomp(pairs[i].Reverse).decode(), a + 150, rng.choice([0, 1]))
        seqs.append("".join(s).encode())
    eng = engine.New(cfg)
    full = eng.CompilePanel(pairs)
    reset = [any(ch not in b"ACGTacgt" for ch in s) for s in seqs]
    mode = 1 if any(reset) else 0
    lens = [len(s) for s in seqs]
    flags = [(1 if reset[i] else 0) | (2 if mode else 0) for i in range(3)]
    whole = np.concatenate([synth_hits(full, s, cfg.MaxMM, i, mode) for i, s in enumerate(seqs)])
    sc = engine.SimulationScratch(full, host_only=True)
    want = [p.sig() for p in eng.JoinHits(full, sc, whole, lens, flags)]
    assert len(want) >= 10
    for count in (2, 3, 7):
        shards = [eng.CompilePanel(pairs) for _ in range(count)]
        for i, sh in enumerate(shards):
            sh.set_shard(i, count)
        lists = [sh.scanned_patterns(mode) for sh in shards]
        assert sorted(sum(lists, [])) == full.scanned_patterns(mode)            # a partition of the pattern list
        assert max(map(len, lists)) - min(map(len, lists)) <= 1                   # balanced
        parts = [np.concatenate([synth_hits(sh, s, cfg.MaxMM, i, mode) for i, s in enumerate(seqs)]) for sh in shards]
        rng.shuffle(parts)
        got = [p.sig() for p in eng.JoinHits(full, sc, np.concatenate(parts), lens, flags)]
        assert got == want, count
        with pytest.raises(_lib.IpcrError):
            shards[0].set_shard(0, 2)                                            # already a shard
        for sh in shards:
            sh.close()
    with pytest.raises(_lib.IpcrError):
        full.set_shard(3, 3)


@pytest.mark.parametrize("k,tw,lens,iupac", [(2, 3, (20, 20), False), (1, 5, (18, 25), False), (3, 3, (22, 30), True),
                                             (2, 0, (20, 24), False), (0, 3, (16, 21), False)])
def test_seed_index_source_compiles_for_gfx950(k, tw, lens, iupac, tmp_path):
    """the seed-index kernel is generated per panel (key shapes, bitmap sizes, queue size all baked in): its
    source for panels of different k / window / primer lengths must compile for gfx950, keep its LDS within the CU's
    160 KiB and its hot registers within four waves per SIMD (hipcc cross-compiles without a GPU)"""
    import subprocess
    rng = random.Random(k * 100 + tw * 10 + lens[0])
    alphabet = "ACGT"
    rows = []
    for i in range(40):
        def mk():
            s = [rng.choice(alphabet) for _ in range(rng.randint(*lens))]
            if iupac and rng.random() < 0.3:
                s[rng.randrange(3, len(s) - 3)] = rng.choice("RYMK")
            return "".join(s)
        rows.append(primer.Pair("r%d" % i, mk(), mk(), 0, 0))
    cp = engine.New(engine.Config(MaxMM=k, TerminalWindow=tw, MaxLen=2000)).CompilePanel(primer.AddSelfPairsUnique(rows))
    src = cp.filter_source(2)
    cp.close()
    assert "ipcr_index_filter" in src
    if "two steps per lookup" in src:                          # 3 protected + 5 block bases throughout: one ds_read_b32 per shape and PAIR of steps
        assert re.search(r"const u32 cp0_1 = \(.*& 60u\), un0_1 = ANDOR\(.*, 3u, 16u\), uo0_1 = ", src), \
            "the protected bases two consecutive steps share must be taken once per pair and group"
        assert len(re.findall(r"reinterpret_cast<const u32\*>\(ldsb \+", src)) == 16 * src.count("#define NS ") * int(re.search(r"#define NS (\d+)", src).group(1))
    elif tw >= 3 and k >= 1 and "(entry layout C)" not in src:   # (primers beyond 26 nt: one bit per shape, see jit.cpp)
        assert re.search(r"const u32 cb0_0 = \(.*& 7u\), cw0_0 = ", src), \
            "a panel with >= 3 protected bases must take its shapes' common six key bits once per step"
    path = tmp_path / "index.hip"
    path.write_text(src)
    asm = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-o", "-", str(path)],
                                  stderr=subprocess.DEVNULL).decode()
    lds = int(re.search(r"\.group_segment_fixed_size: (\d+)", asm).group(1))
    vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", asm).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", asm).group(1))
    assert lds <= 160 * 1024 and vgpr <= 128 and spill <= 16, (lds, vgpr, spill)


PACK_CHECK = r'''
import ctypes as C, random, sys
from ipcr_amd import _lib
L = _lib.lib()
rng = random.Random(3)
CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
def ref(seq, padded):   # the semantics of kernels.hip: pack_pair, one base at a time
    W = padded // 32
    lo, hi, iv, rs, fl = [0] * W, [0] * W, [0] * W, [0] * W, 0
    for i in range(padded):
        w, b = divmod(i, 32)
        if i < len(seq):
            c = chr(seq[i])
            if c in "ACGTacgt":
                lo[w] |= (CODE[c.upper()] & 1) << b
                hi[w] |= (CODE[c.upper()] >> 1) << b
                if c.islower():
                    iv[w] |= 1 << b
                    fl |= 2
            else:
                iv[w] |= 1 << b
                rs[w] |= 1 << b
                fl |= 1
        else:
            iv[w] |= 1 << b          # padding: invalid, not a reset byte
    return [lo, hi, iv, rs], fl
for trial in range(400):
    n = rng.choice([0, 1, 31, 32, 33, 63, 64, 65, 100, 1000, 4097])
    kind = rng.random()
    seq = bytes((ord(rng.choice("ACGT")) if kind < 0.3 else ord(rng.choice("ACGTacgtN")) if kind < 0.6 else rng.randrange(256))
                for _ in range(n))
    padded = ((n + 31) // 32) * 32 + 32 * rng.choice([0, 1, 4])
    W = padded // 32
    arrs = [(C.c_uint32 * max(W, 1))() for _ in range(4)]
    f = C.c_uint32(99)
    _lib.check(L.ipcr_pack_ascii(seq, n, padded, *arrs, C.byref(f)))
    want, fl = ref(seq, padded)
    assert [list(a)[:W] for a in arrs] == want and f.value == fl, (trial, n, padded)
print("ok")
'''


@pytest.mark.parametrize("scalar,avx512", [("0", "1"), ("0", "0"), ("1", "0")])
def test_host_packer_matches_the_pack_kernel_semantics(scalar, avx512):
    """ipcr_pack_ascii (csrc/hostpack.cpp: what ipcr_scan_chunk sends over PCIe instead of ASCII) against a
    base-at-a-time statement of the device pack kernel's semantics: upper-case ACGT valid (core/primer/iupac.go:62-67),
    lower-case acgt invalid but not a reset byte (core/engine/ac.go:16-30), everything else both, padding invalid only;
    AVX-512BW (where the CPU has it), AVX2 and scalar paths, every byte value, lengths around the 32-base word."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", PACK_CHECK], capture_output=True, text=True, cwd=root,
                       env=dict(os.environ, IPCR_PACK_SCALAR=scalar, IPCR_PACK_AVX512=avx512, PYTHONPATH=root))
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
    with pytest.raises(_lib.IpcrError):
        _lib.check(_lib.lib().ipcr_pack_ascii(b"ACGT", 4, 33, None, None, None, None, None))


def test_profiles_readme_is_generated():
    """profiles/README.md is written by tools/collect_profiles.py from the CSV / JSON files beside it: every figure in it is
    a value of the file its row names.  Regenerated here and compared -- prose cannot drift from the evidence."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("collect_profiles", os.path.join(root, "tools", "collect_profiles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = mod.readme_text()
    have = open(os.path.join(root, "profiles", "README.md")).read()
    assert have == want, "profiles/README.md is stale: run `python tools/collect_profiles.py readme`"


def test_large_hit_lists_are_joined_per_record_in_parallel(monkeypatch):
    """join_sorted_hits hands a hit list of 32 768 records and more to the process's pool, one record per item, and emits
    the records' product lists in record order: the same products, in the same order, as the one-thread join
    (IPCR_JOIN_PARALLEL=0), emit callbacks and their abort included."""
    import ctypes as C
    from ipcr_amd.dist import HIT_DTYPE
    rng = np.random.default_rng(12)
    pairs = [primer.Pair("p%d" % i, "ACGTACGTACGTACGTAC"[: 16 + i % 3], "TTGGCCAATTGGCCAATTGG"[: 17 + i % 4], 0, 0) for i in range(6)]
    cfg = engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=400, HitCap=50, SeedLen=12)
    cp = engine.New(cfg).CompilePanel(pairs)
    nrec, per = 40, 1800        # 72 000 hits: the parallel sort (>= 65 536) and the parallel join (>= 32 768)
    hits = np.zeros(nrec * per, dtype=HIT_DTYPE)
    pats = cp.scanned_patterns(0)
    hits["record"] = np.repeat(np.arange(nrec, dtype=np.uint32), per)
    hits["pattern"] = rng.choice(pats, nrec * per).astype(np.uint32)
    hits["pos"] = rng.integers(0, 20_000, nrec * per)
    hits["mm0"] = rng.choice([0, 0, 1 << 9, (1 << 5) | (1 << 11)], nrec * per).astype(np.uint64)
    rng.shuffle(hits)
    assert len(hits) >= 65536
    lens, flags = [30_000] * nrec, [0] * nrec
    sc = engine.SimulationScratch(cp, host_only=True)
    monkeypatch.setenv("IPCR_JOIN_PARALLEL", "0")
    want = [(p.SequenceID,) + p.sig() for p in engine.New(cfg).JoinHits(cp, sc, hits, lens, flags)]
    monkeypatch.setenv("IPCR_JOIN_PARALLEL", "1")
    got = [(p.SequenceID,) + p.sig() for p in engine.New(cfg).JoinHits(cp, sc, hits, lens, flags)]
    assert got == want and len(want) > 1000
    assert [w[0] for w in want] == sorted((w[0] for w in want), key=int)       # record order
    # emit: called in the same order, and a non-zero return stops the hand-out
    L = _lib.lib()
    seen = []
    stop_at = 137

    @_lib.EMIT_FN
    def cb(pp, _u):
        seen.append((pp.contents.record, pp.contents.pair, pp.contents.start, pp.contents.end, pp.contents.type))
        return 1 if len(seen) == stop_at else 0

    lens_c = (C.c_uint64 * nrec)(*lens)
    fl_c = (C.c_uint8 * nrec)(*flags)
    st = L.ipcr_join_hits(cp._h, sc._h, C.c_void_p(hits.ctypes.data), len(hits), lens_c, fl_c, nrec, C.cast(cb, C.c_void_p), None)
    assert st == _lib.ERR_ABORTED and len(seen) == stop_at
    first = [(int(w[0]), w[2], w[3]) for w in want[:stop_at]]
    assert [(r, s, e) for (r, _p, s, e, _t) in seen] == first
    sc.close()
    cp.close()


def test_pack_pool_runs_every_item_exactly_once():
    """the process's pool of pack threads (host.cpp: PackPool -- per-thread mailboxes, a first item by thread number, a shared
    counter for the rest): thousands of runs of 1..200 items from two callers at once, with and without the idle callback of the
    lone-worker chunk path; no item may be skipped or run twice, whichever threads were polling, asleep or late"""
    import ctypes
    from ipcr_amd import _lib
    fn = _lib.lib().ipcr_internal_pool_selftest
    fn.restype = ctypes.c_int32
    fn.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    assert fn(3000, 200) == 0
    assert fn(300, 3) == 0        # fewer items than threads


def test_chunk_windows_are_the_streaming_readers(tmp_path):
    """ipcr_chunk_windows (what ipcr_scan_genome_chunked cuts a resident record into) against the streaming reader's rolling
    chunks (ipcr_fasta_next, itself pinned by the reference's chunk literals in tests/test_fasta_cli.py): record lengths around
    the chunk size, the step and their multiples, overlaps from 0 to chunk - 1, chunking off"""
    import ctypes as C
    import random
    from ipcr_amd import fasta
    rng = random.Random(5)
    cases = [(n, c, o) for c, o in ((10, 3), (10, 0), (7, 6), (16, 5), (100, 20)) for n in (0, 1, c - 1, c, c + 1, 2 * c - o, 2 * c - o + 1, 3 * c, 57)]
    cases += [(rng.randrange(0, 400), c, rng.randrange(0, c)) for c in (5, 11, 64) for _ in range(12)]
    cases += [(25, 0, 0), (25, 10, 10), (25, 10, 12)]        # chunking off: no size, or no step
    for n, chunk, overlap in cases:
        path = tmp_path / "w.fa"
        seq = "".join(rng.choice("ACGT") for _ in range(n))
        path.write_text(">r one\n" + "\n".join(seq[i:i + 13] for i in range(0, n, 13)) + "\n")
        want = [r.ID for r in fasta.StreamChunks(str(path), chunk, overlap)]
        cnt = C.c_int64()
        _lib.check(_lib.lib().ipcr_chunk_windows(n, chunk, overlap, None, 0, C.byref(cnt)))
        w = (_lib.ChunkWindow * max(cnt.value, 1))()
        _lib.check(_lib.lib().ipcr_chunk_windows(n, chunk, overlap, w, cnt.value, C.byref(cnt)))
        got = ["r" if w[i].plain else "r:%d-%d" % (w[i].start, w[i].end) for i in range(cnt.value)]
        assert got == want, (n, chunk, overlap)


def test_fasta_host_packer_equals_the_streaming_reader(tmp_path):
    """FASTA text packed on the host (csrc/fasta_hostpack.cpp: line ends squeezed out of the classification masks with pext, 2 bits
    per base, 64-bit words; the resident loader's fast way in, written through the PCIe BAR on the GPU box) against the streaming
    reader's records packed by pack_linear: IDs, lengths and every word of every plane -- line widths around the word and block
    sizes, "\\n" and "\\r\\n", a last line of any length with or without its end, empty records, headers without an ID, text in
    front of the first header, N runs and lower case.  Files that are not that regular (a short or long line in the middle, a
    blank line, a blank or tab at a line's end or start, a lone CR) must be REFUSED (the device loader takes them), never packed."""
    import ctypes
    import random
    fn = _lib.lib().ipcr_internal_fasta_hostpack_check
    fn.restype = ctypes.c_int32
    fn.argtypes = [ctypes.c_char_p]
    probe = tmp_path / "p.fa"
    probe.write_text(">a\nACGT\n")
    if fn(str(probe).encode()) == -1:
        pytest.skip("no AVX-512BW + BMI2 on this host: the loader takes the device path")
    rng = random.Random(3)
    path = str(tmp_path / "fhp.fa")

    def seq(n, junk=0.0, lower=0.0):
        s = [rng.choice("ACGT") for _ in range(n)]
        for i in range(n):
            x = rng.random()
            if x < junk: s[i] = rng.choice("NRYKMnx-")
            elif x < junk + lower: s[i] = s[i].lower()
        return "".join(s)
    def write(path, recs, W, nl="\n", final_nl=True, lead=""):
        with open(path, "w", newline="") as fh:
            fh.write(lead)
            for k, (hdr, s) in enumerate(recs):
                fh.write(">" + hdr + nl)
                lines = [s[i:i + W] for i in range(0, len(s), W)]
                last = k == len(recs) - 1
                fh.write(nl.join(lines) + (nl if lines and (final_nl or not last) else ""))
    for case in range(150):
        W = rng.choice([1, 7, 31, 32, 60, 61, 63, 64, 65, 70, 80, 100, 127, 128, 129, 1000])
        nl = rng.choice(["\n", "\n", "\r\n"])
        nrec = rng.randint(1, 5)
        recs = []
        for r in range(nrec):
            n = rng.choice([0, 1, W - 1 if W > 1 else 1, W, W + 1, 2 * W, 3 * W + 5, rng.randint(0, 5000), rng.randint(60000, 300000)])
            recs.append(("r%d desc text > more" % r, seq(n, junk=rng.choice([0, 0, 0.001, 0.05]), lower=rng.choice([0, 0.01, 0.3]))))
        if rng.random() < 0.2: recs.insert(rng.randrange(len(recs) + 1), ("", seq(100)))      # header without an ID: dropped
        write(path, recs, W, nl, final_nl=rng.random() < 0.7, lead=rng.choice(["", "", "ACGT" + nl, nl]))
        rc = fn(path.encode())
        assert rc == 0, (case, W, nl, [len(s) for _, s in recs], rc)
    # irregular files must be refused (-1 / -2), never packed wrongly
    for case in range(60):
        W = rng.choice([60, 80])
        s = seq(rng.randint(500, 20000))
        lines = [s[i:i + W] for i in range(0, len(s), W)]
        kind = case % 6
        if kind == 0 and len(lines) > 3: lines[2] = lines[2][:-5]                  # a short line in the middle
        elif kind == 1: lines[1] = lines[1] + " "                                   # trailing blank
        elif kind == 2: lines.insert(2, "")                                         # blank line
        elif kind == 3: lines[1] = "\t" + lines[1][1:]                              # leading tab
        elif kind == 4 and len(lines) > 3: lines[2] = lines[2] + "ACGT"             # a long line
        elif kind == 5: lines[0] = lines[0][:10] + "\r" + lines[0][11:]             # a lone CR inside a line
        open(path, "w", newline="").write(">x\n" + "\n".join(lines) + "\n")
        rc = fn(path.encode())
        assert rc < 0, (case, kind, rc)


def test_null_stream_fills_are_waited_for():
    """Every stream the library creates is non-blocking, and hipMemset(dev, ...) runs on the null stream without waiting for the
    device: a fill of fresh device memory that is followed by work on one of the library's streams can land AFTER that work
    (round 4: the exchange's staging block was cleared after its header had been staged -- a rank reported zero hits).  The
    rule the sources keep: a synchronous-looking hipMemset is followed, before anything else touches the memory, by
    hipStreamSynchronize(nullptr); fills that belong to a stream's order use hipMemsetAsync on that stream."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = 0
    for path in sorted(glob.glob(os.path.join(root, "ipcr_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(root, "ipcr_amd", "csrc", "*.hip"))):
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            if re.search(r"\bhipMemset(D8|D16|D32)?\s*\(", line) and "//" not in line.split("hipMemset")[0]:
                seen += 1
                window = "\n".join(lines[i:i + 12])
                assert "hipStreamSynchronize(nullptr)" in window or "hipDeviceSynchronize()" in window, \
                    "%s:%d: hipMemset on the null stream is not waited for before the library's non-blocking streams go on" % (os.path.basename(path), i + 1)
    assert seen >= 3
