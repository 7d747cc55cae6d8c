"""Device-side FASTA normalisation (ipcr_genome_add_fasta: raw slabs -> fasta_kernels.hip -> pack kernel)
against the line-by-line host reader, which restates core/fasta (scan.go:10-69, normalize.go:5-14,
stream.go:125-131, path_ctx.go:142-144) and is itself pinned by tests/test_fasta_cli.py."""
import gzip
import os
import random

import pytest

pytestmark = pytest.mark.gpu


def as_tiles(seq: bytes) -> bytes:
    """what ipcr_genome_read returns for a normalised record: ACGT stay, everything else reads back as N"""
    return bytes(c if c in b"ACGT" else ord("N") for c in seq)


def load_and_compare(path, slab=None):
    from ipcr_amd import engine, fasta
    want = [(r.ID, r.Seq) for r in fasta.StreamChunks(path)]
    old = os.environ.pop("IPCR_FASTA_SLAB", None)
    if slab is not None:
        os.environ["IPCR_FASTA_SLAB"] = str(slab)
    try:
        g = engine.Genome(max(sum(len(s) for _, s in want) * 2, 1 << 16) + 8192 * (len(want) + 2), max_records=len(want) + 4)
        n = g.add_fasta(path)
    finally:
        os.environ.pop("IPCR_FASTA_SLAB", None)
        if old is not None:
            os.environ["IPCR_FASTA_SLAB"] = old
    assert n == len(want) == g.num_records
    assert g.ids == [i for i, _ in want]
    for r, (_, seq) in enumerate(want):
        assert g.record_len(r) == len(seq)
        assert g.read(r, 0, len(seq)) == as_tiles(seq), (r, slab)
        has_reset = any(c not in b"ACGTacgt" for c in seq)
        assert (g.record_flags(r) & 1) == int(has_reset)
    g.close()
    return want


TRICKY = (
    b"stray line before any header\nACGT\n"
    b">r1 first record\tdesc\nACGTacgtNNnn\n  ACG T  \r\n\n\t\nTTTT\r\n"
    b">  r2   padded header id\n\n\n"
    b">r3\nA\nC\nG\nT\n>\nAAAA dropped: header without id\n>r4|x more\n"
    b"acgtRYKMswbdhvn-*.\nGG>GG not a header\n \x0b\x0c \nCCCC"
    b"\n>r5\nTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"
)


@pytest.mark.parametrize("slab", [None, 64, 80, 96, 128, 4096])
def test_tricky_file_every_slab_size(tmp_path, slab):
    p = tmp_path / "tricky.fa"
    p.write_bytes(TRICKY)
    want = load_and_compare(str(p), slab)
    assert [i for i, _ in want] == ["r1", "r2", "r3", "r4|x", "r5"]
    assert want[0][1] == b"ACGTACGTNNNNACG TTTTT" and want[1][1] == b"" and want[2][1] == b"ACGT"


def test_no_final_newline_header_last_and_empty_file(tmp_path):
    p = tmp_path / "a.fa"
    p.write_bytes(b">x\nACGT\n>y")            # header is the last line, no line end: empty record y
    assert load_and_compare(str(p), 64) == [("x", b"ACGT"), ("y", b"")]
    q = tmp_path / "empty.fa"
    q.write_bytes(b"")
    assert load_and_compare(str(q)) == []
    w = tmp_path / "ws.fa"
    w.write_bytes(b">z\nAC\n   \n \t \n\n")
    assert load_and_compare(str(w), 64) == [("z", b"AC")]


def test_header_line_longer_than_the_slab_is_refused(tmp_path):
    """A header line that does not end inside its slab: the device-side search would clamp it to the slab -- a truncated
    ID, the rest of the line decoded as sequence.  The loader refuses the file (IPCR_ERR_UNSUPPORTED); a header at the very
    end of the file (no line end at all) is still an empty record, and the same file loads with a slab that holds the line."""
    from ipcr_amd import _lib, engine
    p = tmp_path / "longhdr.fa"
    hdr = b">id " + b"x" * 300
    p.write_bytes(b">a\nACGT\n" + hdr + b"\nGGGG\n")
    os.environ["IPCR_FASTA_SLAB"] = "128"
    try:
        g = engine.Genome(1 << 16, max_records=8)
        with pytest.raises(_lib.IpcrError) as ei:
            g.add_fasta(str(p))
        assert ei.value.status == _lib.ERR_UNSUPPORTED and "header line longer" in ei.value.message
        g.close()
    finally:
        os.environ.pop("IPCR_FASTA_SLAB", None)
    assert load_and_compare(str(p), 1024) == [("a", b"ACGT"), ("id", b"GGGG")]
    q = tmp_path / "hdr_at_eof.fa"
    q.write_bytes(b">a\nACGT\n" + hdr)                 # the last line is a long header without a line end
    assert load_and_compare(str(q), 4096) == [("a", b"ACGT"), ("id", b"")]


def test_long_lines_span_slabs(tmp_path):
    """one sequence line much longer than the slab, with blanks inside and at both ends"""
    rng = random.Random(5)
    body = bytes(rng.choice(b"ACGTacgtN") for _ in range(3000))
    line = b"  \t" + body[:1000] + b" " + body[1000:2000] + b"\t \t" + body[2000:] + b"   \r\n"
    p = tmp_path / "long.fa"
    p.write_bytes(b">long\n" + line + b">next\nACGT\n")
    for slab in (64, 256, 1024, None):
        want = load_and_compare(str(p), slab)
        assert len(want[0][1]) == 3004 and want[1] == ("next", b"ACGT")


def test_random_files_gzip_and_plain(tmp_path):
    rng = random.Random(11)
    for case in range(6):
        parts = []
        for r in range(rng.randint(1, 12)):
            parts.append(b">rec%d some text\n" % r if rng.random() < 0.9 else b">\n")
            n = rng.choice([0, 1, 59, 60, 61, 500, 5000, 70000])
            seq = bytes(rng.choice(b"ACGTACGTACGTacgtNnRY") for _ in range(n))
            width = rng.choice([1, 7, 60, 80, 10 ** 9])
            eol = rng.choice([b"\n", b"\r\n"])
            for i in range(0, n, width):
                parts.append(seq[i:i + width] + eol)
            if rng.random() < 0.3:
                parts.append(eol)
        raw = b"".join(parts)
        p = tmp_path / ("r%d.fa" % case)
        p.write_bytes(raw)
        load_and_compare(str(p), rng.choice([64, 1024, 65536, None]))
        z = tmp_path / ("r%d.fa.gz" % case)
        with gzip.open(z, "wb") as fh:
            fh.write(raw)
        load_and_compare(str(z), 4096)


def test_cli_uses_record_ids_from_loader(tmp_path):
    import io
    from ipcr_amd import cli
    p = tmp_path / "g.fa"
    p.write_text(">chrA desc\nTTTTACGTACGTACGTTTTTGGGGCCCCAAAATTTT\n>chrB\nacgtacgtaaaa\n")
    out = io.StringIO()
    rc = cli.run(["-f", "ACGTACGT", "-r", "GGGGCCCC", "--mismatches", "0", "--max-length", "100", str(p)], stdout=out)
    rows = out.getvalue().splitlines()
    assert rc == 0 and rows[0] == cli.TSV_HEADER
    assert any(row.split("\t")[1] == "chrA" for row in rows[1:])


def _cli(args):
    import io
    from ipcr_amd import cli
    out, err = io.StringIO(), io.StringIO()
    rc = cli.run(args, stdout=out, stderr=err)
    assert rc == 0, err.getvalue()
    return out.getvalue(), err.getvalue()


@pytest.mark.parametrize("stream", ["", "1"])
def test_chunking_keeps_boundary_hits(tmp_path, monkeypatch, stream):  # internal/integration/integration_test.go:125-225
    """--chunk-size output == unchunked output: rolling chunks (the resident genome scanned in windows, or every window through
    ipcr_scan_chunk), collector rebasing and de-duplication (internal/pipeline/pipeline.go:127-161) against the resident
    whole-record scan"""
    monkeypatch.setenv("IPCR_CLI_STREAM_CHUNKS", stream)
    fa = tmp_path / "chunk.fa"
    fa.write_text(">s\nACGTACGTACGTACGTACGTACGTACGT\n")
    base = ["--forward", "ACGTAC", "--reverse", "ACGTAC", "--sort", "--max-length", "8"]
    whole, _ = _cli(base + [str(fa)])
    chunked, _ = _cli(base + ["--chunk-size", "16", str(fa)])
    assert whole == chunked and len(whole.splitlines()) > 3
    # chunk-size <= max product length disables chunking with the reference's warning (runutil.go:54-57)
    same, warn = _cli(base + ["--chunk-size", "8", str(fa)])
    assert same == whole and "disabling chunking" in warn


@pytest.mark.parametrize("stream", ["", "1"])   # "": the resident genome scanned in rolling windows (one sweep); "1": every window through ipcr_scan_chunk
def test_chunked_equals_unchunked_on_planted_records(tmp_path, monkeypatch, stream):
    import ipcr_oracle as O
    monkeypatch.setenv("IPCR_CLI_STREAM_CHUNKS", stream)
    rng = random.Random(91)
    fwd, rev = "ACGTTGCATGCAAGCTTA", "GGCCTTAAGGCCATATCG"
    rc = O.revcomp(rev).decode()
    recs = []
    for r in range(3):
        s = list(O.bench_dna(60000 + 777 * r, 500 + r).decode())
        for t in range(12):                     # amplicons everywhere, many across chunk boundaries
            a = 300 + t * 4900 + rng.randrange(50)
            ln = rng.choice([150, 400, 900])
            s[a:a + len(fwd)] = fwd
            s[a + ln - len(rc):a + ln] = rc
        if r == 1:
            s[20000:20040] = "N" * 40
        recs.append(("ctg%d" % r, "".join(s)))
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">%s desc\n%s\n" % (i, "\n".join(s[j:j + 60] for j in range(0, len(s), 60))) for i, s in recs))
    base = ["-f", fwd, "-r", rev, "-m", "1", "--sort", "--max-length", "1000"]
    whole, _ = _cli(base + [str(fa)])
    assert len(whole.splitlines()) >= 30
    for cs in (5000, 7777, 20000):
        chunked, _ = _cli(base + ["--chunk-size", str(cs), str(fa)])
        assert chunked == whole, cs


def test_thousands_of_tiny_records_batched(tmp_path):
    """a fragmented assembly: records that begin and end inside one slab are packed by ONE launch straight out of the
    compacted slab, at whatever byte they start; records cut by a slab edge take the record-buffer path"""
    rng = random.Random(2024)
    parts = []
    for r in range(3000):
        n = rng.choice([0, 1, 15, 16, 17, 59, 60, 61, 128, 200, 1000])
        seq = bytes(rng.choice(b"ACGTACGTACGTacgtN") for _ in range(n))
        parts.append(b">c%d len=%d\n" % (r, n))
        for i in range(0, n, 60):
            parts.append(seq[i:i + 60] + b"\n")
    p = tmp_path / "frag.fa"
    p.write_bytes(b"".join(parts))
    for slab in (None, 65536, 4096):
        load_and_compare(str(p), slab)


def test_regular_files_are_packed_on_the_host_and_written_through_the_bar(tmp_path):
    """ipcr_genome_add_fasta's fast way in (csrc/fasta_hostpack.cpp + host.cpp: the text packed on the host, line ends squeezed out
    with pext, the code planes written straight into device memory through the PCIe BAR, the invalid-bit plane only for groups of
    columns that hold an invalid base): regular files of several shapes against the streaming reader -- IDs, lengths, every base
    read back from the tiles, the reset flags -- and the counter says that this loader took them; an irregular file, a gzip file
    and a file under IPCR_FASTA_SLAB take the device loader and give the same records."""
    import ctypes
    from ipcr_amd import _lib, engine, fasta
    loads = _lib.lib().ipcr_internal_fasta_hostpacked_loads
    loads.restype = ctypes.c_uint64
    how = _lib.lib().ipcr_internal_device_bar
    how.restype, how.argtypes = ctypes.c_int32, [ctypes.c_int32]
    rng = random.Random(77)

    def seq(n, junk, lower):
        s = [rng.choice("ACGT") for _ in range(n)]
        for i in range(n):
            x = rng.random()
            if x < junk:
                s[i] = rng.choice("NRYKMnx-")
            elif x < junk + lower:
                s[i] = s[i].lower()
        return "".join(s)

    to_n = bytes(c if c in b"ACGT" else ord("N") for c in range(256))       # what is not a base reads back as one invalid code

    def check(path, expect_fast):
        before = loads()
        g = engine.Genome(40_000_000, max_records=64)
        g.add_fasta(str(path))
        want = list(fasta.StreamChunks(str(path), 0, 0))
        assert g.ids == [r.ID for r in want]
        for i, r in enumerate(want):
            assert g.record_len(i) == len(r.Seq)
            # the tiles keep the case-folded text: every base is read back (N and other codes come back as 'N'? no: as what was packed)
            got = g.read(i, 0, len(r.Seq)) if len(r.Seq) else b""
            assert got.translate(to_n) == r.Seq.translate(to_n), (str(path), i)
            assert (g.record_flags(i) & 1) == (1 if r.Seq.translate(None, b"ACGTacgt") else 0)
        g.close()
        if expect_fast is not None and how(0) == 2:
            assert (loads() - before == 1) == expect_fast, (str(path), expect_fast)

    for k, (W, nl, final_nl) in enumerate(((60, "\n", True), (80, "\r\n", True), (61, "\n", False), (1000, "\n", True), (7, "\n", True))):
        p = tmp_path / ("reg%d.fa" % k)
        with open(p, "w", newline="") as fh:
            fh.write("text in front of the first header" + nl)
            for r, n in enumerate((9_000_001, 0, W, 3 * W, 123_457, 5)):
                s = seq(n if W > 7 else min(n, 50_000), rng.choice([0, 0.001]), rng.choice([0, 0.2]))
                if r == 0 and W == 60:
                    s = s[:4_000_000] + "N" * 5000 + s[4_005_000:]     # a run of N in ONE group of columns: only its plane follows
                fh.write(">rec%d description > text%s" % (r, nl))
                if r == 3:
                    fh.write(">%sACGT%s" % (nl, nl))                     # a header without an ID drops its record
                lines = [s[i:i + W] for i in range(0, len(s), W)]
                fh.write(nl.join(lines) + (nl if lines and (final_nl or r < 5) else ""))
        check(p, True)
    irr = tmp_path / "irregular.fa"
    irr.write_text(">a\nACGTACGT\nACG\nACGTACGT\n>b\nACGT \nAC\n")
    check(irr, False)
    gz = tmp_path / "reg.fa.gz"
    with gzip.open(gz, "wt") as fh:
        fh.write(">z\n" + "\n".join(["ACGTACGTAC"] * 50) + "\n")
    check(gz, False)
    os.environ["IPCR_FASTA_SLAB"] = "4096"
    try:
        check(tmp_path / "reg0.fa", False)
    finally:
        del os.environ["IPCR_FASTA_SLAB"]
