"""FASTA record/rolling-chunk stream (core/fasta) and the CLI-side formatting helpers: CPU tests;
the end-to-end CLI runs (config C1 and friends) need a GPU."""
import gzip
import io
import os

import pytest

import ipcr_oracle as O
from ipcr_amd import cli, fasta, engine

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def chunks(path, cs=0, ov=0):
    return [(r.ID, r.Seq) for r in fasta.StreamChunks(path, cs, ov)]


def test_rolling_overlap(tmp_path):  # core/fasta/reader_test.go:91-116
    p = tmp_path / "chunk.fa"
    p.write_text(">s\nACGTACGTACGT\n")
    assert chunks(str(p), 5, 2) == [("s:0-5", b"ACGTA"), ("s:3-8", b"TACGT"), ("s:6-11", b"GTACG"), ("s:9-12", b"CGT")]


def test_short_record_keeps_base_id(tmp_path):  # reader_test.go:118-135
    p = tmp_path / "short.fa"
    p.write_text(">s\nACGTA\n")
    assert chunks(str(p), 5, 2) == [("s", b"ACGTA")]


def test_gzip_and_normalisation(tmp_path):  # reader_test.go:15-66, normalize.go:5-14
    p = tmp_path / "t.fa.gz"
    with gzip.open(p, "wt") as fh:
        fh.write(">seq1\nACGT\n>seq2\nNNnn\n")
    assert chunks(str(p)) == [("seq1", b"ACGT"), ("seq2", b"NNNN")]
    q = tmp_path / "noext"  # gzip detected by magic, not by suffix (open.go:40-42)
    q.write_bytes(p.read_bytes())
    assert chunks(str(q)) == [("seq1", b"ACGT"), ("seq2", b"NNNN")]


def test_header_id_and_stray_lines(tmp_path):  # stream.go:125-131, path_ctx.go:142-144
    p = tmp_path / "w.fa"
    p.write_bytes(b"ACGT\n>a\tb c\n ac gt \r\nAC\n\n>empty\n>  lead trail  \nGG")
    assert chunks(str(p)) == [("a", b"AC GTAC"), ("empty", b""), ("lead", b"GG")]


def test_chunk_schedule_matches_reference_rule(tmp_path):  # path_ctx.go:146-163 on many shapes
    import random
    rng = random.Random(3)
    for _ in range(40):
        n, cs, ov, width = rng.randint(0, 400), rng.randint(1, 60), rng.randint(0, 70), rng.choice([7, 60, 1000])
        seq = "".join(rng.choice("ACGTN") for _ in range(n))
        p = tmp_path / "r.fa"
        p.write_text(">id x\n" + "\n".join(seq[i:i + width] for i in range(0, n, width)) + ("\n" if n else ""))
        got = chunks(str(p), cs, ov)
        step = cs - ov
        if step <= 0 or n <= cs:
            want = [("id", seq.encode())]
        else:
            want, ws, last = [], 0, 0
            while n - ws > cs:
                want.append(("id:%d-%d" % (ws, ws + cs), seq[ws:ws + cs].encode()))
                last = ws + cs
                ws += step
            if last < n:
                want.append(("id:%d-%d" % (ws, n), seq[ws:n].encode()))
        assert got == want, (n, cs, ov, width)


def test_split_chunk_suffix():  # internal/common/ids.go:11-27
    assert cli.split_chunk_suffix("chr1:100-200") == ("chr1", 100, True)
    assert cli.split_chunk_suffix("chr1") == ("chr1", 0, False)
    assert cli.split_chunk_suffix("a:b:7-9") == ("a:b", 7, True)
    assert cli.split_chunk_suffix("x:") == ("x:", 0, False)
    assert cli.split_chunk_suffix("x:ab-3") == ("x:ab-3", 0, False)
    # strconv.Atoi (ids.go:21), not Python's int(): no blanks, no '_' separators, no non-ASCII digits, int64 range
    for odd in ("x: 7-9", "x:1_0-2", "x:7 -9", "x:\u0667-9", "x:-9", "x:--7-9", "x:99999999999999999999-3", "x:+-1-2"):
        assert cli.split_chunk_suffix(odd) == (odd, 0, False), odd
    assert cli.split_chunk_suffix("x:+7-9") == ("x", 7, True)
    assert cli.split_chunk_suffix("x:007-9") == ("x", 7, True)
    assert cli.split_chunk_suffix("x:9223372036854775807-1") == ("x", 9223372036854775807, True)
    assert cli.split_chunk_suffix("x:9223372036854775808-1") == ("x:9223372036854775808-1", 0, False)


def test_row_format_and_order():  # internal/output/rows.go:21-29, internal/common/sort.go:34-78
    P = engine.Product
    a = P("e", "chr1", 10, 30, 20, "forward", 1, 0, (3,), ())
    b = P("e", "chr1", 10, 30, 20, "revcomp", 0, 2, (), (9, 4))
    c = P("e", "chr0:5-100", 2, 22, 20, "forward", 0, 0, (), ())
    assert cli.format_row("g.fa", b) == "g.fa\tchr1\te\t10\t30\t20\trevcomp\t0\t2\t\t9,4"
    rows = sorted([b, a, c], key=lambda p: cli.product_sort_key("g.fa", p))
    assert rows == [c, a, b]  # base id, then global start; 'forward' < 'revcomp'


def test_primer_tsv_loader(tmp_path):  # core/primer/loader.go:11-61
    p = tmp_path / "p.tsv"
    p.write_text("# comment\n\nid1 acgt TTGA\nid2\tACGT\tGGCC\t100\nid3 ACGT GGCC 100 200\n")
    got = cli.load_tsv(str(p))
    assert [(x.ID, x.Forward, x.Reverse, x.MinProduct, x.MaxProduct) for x in got] == [
        ("id1", "ACGT", "TTGA", 0, 0), ("id2", "ACGT", "GGCC", 100, 0), ("id3", "ACGT", "GGCC", 100, 200)]
    p.write_text("only two\n")
    with pytest.raises(ValueError):
        cli.load_tsv(str(p))


@pytest.mark.gpu
def test_config_c1_demo_fa_header_only():
    """BASELINE.json configs[0]: ipcr 27F/1492R --mismatches 0 on demo.fa -> header only, exit 0."""
    out = io.StringIO()
    rc = cli.run(["-f", "AGAGTTTGATCMTGGCTCAG", "-r", "TACGGYTACCTTGTTAYGACTT", "--mismatches", "0",
                  os.path.join(GOLDEN, "demo.fa")], stdout=out)
    assert rc == 0 and out.getvalue() == cli.TSV_HEADER + "\n"


@pytest.mark.gpu
def test_cli_end_to_end_matches_oracle(tmp_path, monkeypatch):
    import random
    rng = random.Random(8)
    recs = []
    fwd, rev = "ACGTTGCATGCAAGCT", "GGCCTTAAGGCCATAT"
    for r in range(3):
        s = list(O.bench_dna(20000, 77 + r).decode())
        for t in range(3):
            a = 500 + t * 5000
            s[a:a + len(fwd)] = fwd
            rc = O.revcomp(rev).decode()
            s[a + 200:a + 200 + len(rc)] = rc
        recs.append(("ctg%d" % r, "".join(s)))
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">%s some description\n%s\n" % (i, "\n".join(s[j:j + 70].lower() for j in range(0, len(s), 70))) for i, s in recs))
    out = io.StringIO()
    rc = cli.run(["-f", fwd, "-r", rev, "-m", "1", "--sort", str(fa)], stdout=out)
    assert rc == 0
    lines = out.getvalue().splitlines()
    assert lines[0] == cli.TSV_HEADER
    cfg = O.Config(max_mm=1, terminal_window=3, max_len=2000, hit_cap=10000, seed_len=12)
    pairs = [O.Pair("manual", fwd, rev, 0, 2000), O.Pair("manual+A:self", fwd, fwd), O.Pair("manual+B:self", rev, rev)]
    want = []
    for i, s in recs:
        for w in O.simulate_batch(cfg, s, pairs):
            want.append(engine.Product(w.experiment_id, i, w.start, w.end, w.length, w.type, w.fwd_mm, w.rev_mm, w.fwd_idx, w.rev_idx))
    want.sort(key=lambda p: cli.product_sort_key(str(fa), p))
    assert lines[1:] == [cli.format_row(str(fa), p) for p in want] and len(want) >= 9
    # ipcr-probe overlay
    out = io.StringIO()
    probe = recs[0][1][600:621]
    rc = cli.run(["-f", fwd, "-r", rev, "-m", "1", "--sort", "--probe", probe, str(fa)], stdout=out)
    lines = out.getvalue().splitlines()
    assert rc == 0 and lines[0] == cli.TSV_HEADER_PROBE and len(lines) >= 2
    assert all(l.split("\t")[13] == "true" for l in lines[1:])
    # ipcr-probe keeps --chunk-size (internal/probeapp/app.go:108): chunked output == unchunked output, annotation
    # columns included (integration_test.go:125-225 makes the claim for ipcr; every chunk's products are annotated
    # from the chunk's own tiles by ipcr_probe_scratch_products)
    for extra in ([], ["--no-require-probe"], ["--probe-max-mm", "2"]):
        a, b = io.StringIO(), io.StringIO()
        prb = probe if not extra or extra[0] != "--probe-max-mm" else probe[:8] + ("A" if probe[8] != "A" else "C") + probe[9:]
        base = ["-f", fwd, "-r", rev, "-m", "1", "--sort", "--probe", prb] + extra
        assert cli.run(base + [str(fa)], stdout=a) == 0
        # both chunked forms: the resident genome scanned in rolling windows (ipcr_scan_genome_chunked: one sweep; the default), and
        # the windows streamed through ipcr_scan_chunk one by one as a worker of the Go pipeline sends them
        for stream in ("", "1"):
            monkeypatch.setenv("IPCR_CLI_STREAM_CHUNKS", stream)
            b, err = io.StringIO(), io.StringIO()
            assert cli.run(base + ["--chunk-size", "3000", str(fa)], stdout=b, stderr=err) == 0
            assert "chunking disabled" not in err.getvalue()
            assert a.getvalue() == b.getvalue() and len(a.getvalue().splitlines()) >= 2, stream
        monkeypatch.delenv("IPCR_CLI_STREAM_CHUNKS")
        rows = [l.split("\t") for l in a.getvalue().splitlines()[1:]]
        for r in rows:                                                   # and both equal oligo.BestHit on the amplicon
            rec = dict(recs)[r[1]]
            w = O.best_hit(rec[int(r[3]):int(r[4])], prb, 2 if extra and extra[0] == "--probe-max-mm" else 0)
            assert r[13] == ("true" if w.found else "false")
            if w.found:
                assert (r[14], r[15], r[16], r[17]) == (w.strand, str(w.pos), str(w.mm), w.site)


def test_chunking_rules():  # internal/runutil/runutil_test.go:20-59
    assert cli.compute_overlap(100, 21) == 100 and cli.compute_overlap(0, 21) == 20
    assert cli.validate_chunking(False, 0, 500, 25) == (0, 0, [])
    for args in ((True, 1000, 500, 25), (False, 1000, 0, 25), (False, 500, 500, 25)):
        cs, ov, warns = cli.validate_chunking(*args)
        assert (cs, ov) == (0, 0) and len(warns) == 1
    assert cli.validate_chunking(False, 2000, 500, 25) == (2000, 500, [])


def test_sort_uses_source_and_global_chunk_coords():  # internal/common/sort_score_test.go:24-42
    P = engine.Product
    ps = [("b.fa", P("x", "s", 0, 8, 8, "forward", 0, 0, (), ())),
          ("a.fa", P("x", "s:10-20", 2, 8, 6, "forward", 0, 0, (), ())),
          ("a.fa", P("x", "s", 4, 10, 6, "forward", 0, 0, (), ()))]
    ps.sort(key=lambda t: cli.product_sort_key(t[0], t[1]))
    assert [(f, p.SequenceID, p.Start) for f, p in ps] == [("a.fa", "s", 4), ("a.fa", "s:10-20", 2), ("b.fa", "s", 0)]


def test_collector_rebases_and_dedups():  # internal/pipeline/pipeline.go:127-161, runutil/lru_set.go
    P = engine.Product
    c = cli.Collector(cap=2)
    a = c.add("f.fa", P("x", "s:100-200", 5, 25, 20, "forward", 0, 0, (), ()))
    assert (a.SequenceID, a.Start, a.End) == ("s", 105, 125)
    assert c.add("f.fa", P("x", "s:90-190", 15, 35, 20, "forward", 0, 0, (), ())) is None      # same global product
    assert c.add("g.fa", P("x", "s:90-190", 15, 35, 20, "forward", 0, 0, (), ())) is not None  # other file
    b = c.add("f.fa", P("x", "plain", 1, 9, 8, "revcomp", 0, 0, (), ()))
    assert (b.SequenceID, b.Start) == ("plain", 1)
    # capacity 2: the first key has been evicted by now and is accepted again (bounded de-dup, as in the reference)
    assert c.add("f.fa", P("x", "s:100-200", 5, 25, 20, "forward", 0, 0, (), ())) is not None


def test_jsonl_rows():  # pkg/api/products_v1.go:6-25 through encoding/json (internal/jsonlutil/jsonlutil.go)
    import json
    P = engine.Product
    a = P("x", "s:0-4", 0, 4, 4, "forward", 0, 0, (), ())
    assert cli.format_jsonl("", a) == '{"experiment_id":"x","sequence_id":"s:0-4","start":0,"end":4,"length":4,"type":"forward"}'
    b = P("a<b", "chr1", 10, 30, 20, "revcomp", 1, 2, (3,), (9, 4))
    line = cli.format_jsonl("g&h.fa", b)
    assert line == ('{"experiment_id":"a\\u003cb","sequence_id":"chr1","start":10,"end":30,"length":20,"type":"revcomp",'
                    '"fwd_mm":1,"rev_mm":2,"fwd_mm_i":[3],"rev_mm_i":[9,4],"source_file":"g\\u0026h.fa"}')
    assert json.loads(line)["experiment_id"] == "a<b" and json.loads(line)["source_file"] == "g&h.fa"
