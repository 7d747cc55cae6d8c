"""Minimal `ipcr` / `ipcr-probe` / `ipcr-multiplex` driver over the HIP engine.

Only what sits directly either side of the scan path ("next" 1 and 2):
FASTA -> resident tiles, the collector's total product order, and the text/TSV rows -- with the
reference's flag names and defaults (internal/clibase/common.go:61-110) so outputs can be
diffed against `ipcr`.  Thermo scoring, pretty blocks, JSON and nested PCR are out of scope.

    python -m ipcr_amd.cli -f AGAGTTTGATCMTGGCTCAG -r TACGGYTACCTTGTTAYGACTT --mismatches 0 demo.fa
"""
from __future__ import annotations

import argparse
import ctypes as C
import functools
import os
import sys
from typing import List, Optional, Sequence

from . import _lib, engine, primer

TSV_HEADER = ("source_file\tsequence_id\texperiment_id\tstart\tend\tlength\ttype\tfwd_mm\trev_mm"
              "\tfwd_mm_i\trev_mm_i")                                   # internal/output/common.go:5
TSV_HEADER_PROBE = TSV_HEADER + ("\tprobe_name\tprobe_seq\tprobe_found\tprobe_strand\tprobe_pos"
                                 "\tprobe_mm\tprobe_site")              # internal/probeoutput/types.go:24-26


def load_tsv(path: str) -> List[primer.Pair]:
    """primer.LoadTSV -- core/primer/loader.go:11-61"""
    out = []
    with open(path) as fh:
        for ln, line in enumerate(fh, 1):
            line = line.strip()
            if not line or line[0] == "#":
                continue
            f = line.split()
            if len(f) < 3 or len(f) > 5:
                raise ValueError(f"{path}:{ln} bad field count")
            p = primer.Pair(f[0], primer.Validate(f[1]), primer.Validate(f[2]))
            if len(f) >= 4:
                p.MinProduct = int(f[3])
            if len(f) == 5:
                p.MaxProduct = int(f[4])
            out.append(p)
    return out


def ints_csv(a: Sequence[int]) -> str:  # internal/output/rows.go:10-19
    return ",".join(str(v) for v in a)


def split_chunk_suffix(seq_id: str):
    """common.SplitChunkSuffix -- internal/common/ids.go:11-27"""
    colon = seq_id.rfind(":")
    if colon == -1 or colon == len(seq_id) - 1:
        return seq_id, 0, False
    suffix = seq_id[colon + 1:]
    dash = suffix.find("-")
    if dash == -1:
        return seq_id, 0, False
    start = _atoi(suffix[:dash])
    if start is None:
        return seq_id, 0, False
    return seq_id[:colon], start, True


def _atoi(text: str):
    """strconv.Atoi (ids.go:21): optional sign, then ASCII digits only, value inside int64 -- Python's int() also
    takes surrounding blanks, '_' separators and non-ASCII digits, which would rebase IDs such as "x: 7-9" or
    "x:1_0-5" that the reference leaves alone.  None where Atoi returns an error."""
    body = text[1:] if text[:1] in ("+", "-") else text
    if not body or not all("0" <= ch <= "9" for ch in body):
        return None
    v = int(body)
    if text[:1] == "-":
        v = -v
    if v < -(1 << 63) or v > (1 << 63) - 1:
        return None
    return v


def compute_overlap(max_len: int, max_primer_len: int) -> int:
    """runutil.ComputeOverlap -- internal/runutil/runutil.go:20-31"""
    if max_len > 0:
        return max_len
    return max(max_primer_len - 1, 0)


def validate_chunking(circular: bool, chunk_size: int, max_len: int, max_primer_len: int):
    """runutil.ValidateChunking -- internal/runutil/runutil.go:33-62: (chunk, overlap, warnings)"""
    if chunk_size <= 0:
        return 0, 0, []
    if circular:
        return 0, 0, ["chunking disabled for circular templates"]
    if max_len <= 0:
        return 0, 0, ["chunking disabled: a finite effective max product length is required to compute safe overlap"]
    if chunk_size <= max_len:
        return 0, 0, [f"chunk-size ({chunk_size}) <= effective max product length ({max_len}): disabling chunking"]
    return chunk_size, compute_overlap(max_len, max_primer_len), []


class Collector:
    """The pipeline's collector goroutine -- internal/pipeline/pipeline.go:127-161: chunk-local coordinates become
    record-global ones (any ID ending ':<int>-...' counts as a chunk, ids.go:11-27), products seen in the overlap
    of two chunks are dropped by a bounded FIFO/LRU set (runutil/lru_set.go, default 200 000 keys)."""

    def __init__(self, cap: int = 0):
        from collections import OrderedDict
        self.cap = cap if cap > 0 else 200_000
        self.seen = OrderedDict()

    def add(self, source_file: str, p: engine.Product):
        base, off, ok = split_chunk_suffix(p.SequenceID)
        if not ok:
            base, off = p.SequenceID, 0
        gs, ge = p.Start + off, p.End + off
        k = (base, source_file, gs, ge, p.Type, p.ExperimentID)
        if k in self.seen:
            self.seen.move_to_end(k, last=False)
            return None
        self.seen[k] = True
        self.seen.move_to_end(k, last=False)
        if len(self.seen) > self.cap:
            self.seen.popitem(last=True)
        if ok:
            p.SequenceID, p.Start, p.End = base, gs, ge
        return p


def product_sort_key(source_file: str, p: engine.Product):
    """common.LessProduct -- internal/common/sort.go:34-78 as a sort key."""
    base, off, ok = split_chunk_suffix(p.SequenceID)
    if not ok:
        base, off = p.SequenceID, 0
    return (source_file, base, p.Start + off, p.End + off, p.Length, p.Type, p.ExperimentID, p.FwdMM, p.RevMM,
            ints_csv(p.FwdMismatchIdx), ints_csv(p.RevMismatchIdx), p.SequenceID)


def format_row(source_file: str, p: engine.Product) -> str:
    """output.FormatBaseRowTSV -- internal/output/rows.go:21-29"""
    return "\t".join([source_file, p.SequenceID, p.ExperimentID, str(p.Start), str(p.End), str(p.Length), p.Type,
                      str(p.FwdMM), str(p.RevMM), ints_csv(p.FwdMismatchIdx), ints_csv(p.RevMismatchIdx)])


def format_jsonl(source_file: str, p: engine.Product) -> str:
    """One line of --output jsonl: api.ProductV1 (pkg/api/products_v1.go:6-25) as encoding/json writes it -- field
    order of the struct, zero / empty `omitempty` fields left out, compact separators, <, > and & escaped."""
    import json
    d = {"experiment_id": p.ExperimentID, "sequence_id": p.SequenceID, "start": p.Start, "end": p.End,
         "length": p.Length, "type": p.Type}
    if p.FwdMM:
        d["fwd_mm"] = p.FwdMM
    if p.RevMM:
        d["rev_mm"] = p.RevMM
    if p.FwdMismatchIdx:
        d["fwd_mm_i"] = list(p.FwdMismatchIdx)
    if p.RevMismatchIdx:
        d["rev_mm_i"] = list(p.RevMismatchIdx)
    if source_file:
        d["source_file"] = source_file
    text = json.dumps(d, separators=(",", ":"), ensure_ascii=False)
    return text.replace("<", "\\u003c").replace(">", "\\u003e").replace("&", "\\u0026")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="ipcr-hip", add_help=True)
    ap.add_argument("--primers", "-p", default="")
    ap.add_argument("--forward", "-f", default="")
    ap.add_argument("--reverse", "-r", default="")
    ap.add_argument("--sequences", "-s", action="append", default=[])
    ap.add_argument("--mismatches", "-m", type=int, default=0)
    ap.add_argument("--min-length", type=int, default=0)
    ap.add_argument("--max-length", type=int, default=2000)
    ap.add_argument("--hit-cap", type=int, default=10000)
    ap.add_argument("--terminal-window", type=int, default=3)
    ap.add_argument("--self", dest="self_", action=argparse.BooleanOptionalAction, default=True)
    ap.add_argument("--seed-length", type=int, default=12)
    ap.add_argument("--circular", "-c", action="store_true")
    ap.add_argument("--sort", action="store_true")
    ap.add_argument("--output", "-o", default="text", choices=["text", "jsonl"])
    ap.add_argument("--no-header", action="store_true")
    ap.add_argument("--multiplex", action="store_true", help="ipcr-multiplex self-pair rule (unique oligos)")
    ap.add_argument("--probe", "-P", default="")
    ap.add_argument("--probe-name", default="probe")
    ap.add_argument("--probe-max-mm", "-M", type=int, default=0)
    ap.add_argument("--require-probe", action=argparse.BooleanOptionalAction, default=True)
    ap.add_argument("--no-match-exit-code", type=int, default=0)
    ap.add_argument("--chunk-size", type=int, default=0, help="scan rolling chunks through ipcr_scan_chunk (0 = whole records resident)")
    ap.add_argument("--dedup-cap", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("fasta", nargs="*")
    return ap


def run(argv: Optional[Sequence[str]] = None, stdout=None, stderr=None) -> int:
    """app.RunContext -- internal/app/app.go:23-114 (scan-relevant part)."""
    stdout = stdout or sys.stdout
    stderr = stderr or sys.stderr
    o = build_parser().parse_args(argv)
    seq_files = list(o.sequences) + list(o.fasta)
    try:
        if o.primers:
            pairs = load_tsv(o.primers)
        else:
            if not o.forward or not o.reverse:
                print("error: --forward and --reverse (or --primers) are required", file=stderr)
                return 2
            pairs = [primer.Pair("manual", primer.Validate(o.forward), primer.Validate(o.reverse),
                                 o.min_length, o.max_length)]              # app.go:98
        if o.self_:
            pairs = primer.AddSelfPairsUnique(pairs) if o.multiplex else primer.AddSelfPairs(pairs)
    except (ValueError, OSError) as e:
        print(f"error: {e}", file=stderr)
        return 2
    if not seq_files:
        print("error: no FASTA input", file=stderr)
        return 2
    tw = o.terminal_window if o.terminal_window >= 1 else 0            # runutil.EffectiveTerminalWindow
    cfg = engine.Config(MaxMM=o.mismatches, TerminalWindow=tw, MinLen=o.min_length, MaxLen=o.max_length,
                        HitCap=o.hit_cap, SeedLen=o.seed_length, Circular=o.circular)
    _lib.check(_lib.lib().ipcr_set_device(o.device))
    eng = engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)
    rows = []
    max_primer_len = max((max(len(p.Forward), len(p.Reverse)) for p in pairs), default=0)
    chunk, overlap, warns = validate_chunking(o.circular, o.chunk_size, cfg.MaxLen, max_primer_len)
    for w in warns:
        print(f"warning: {w}", file=stderr)
    collector = Collector(o.dedup_cap)
    for path in seq_files:
        if chunk and not os.environ.get("IPCR_CLI_STREAM_CHUNKS"):
            # --chunk-size from a resident genome: the file goes through the device loader, the tiles are swept ONCE, and
            # every rolling window is joined as its own ForEachCompiledProduct call (ipcr_scan_genome_chunked) -- the same
            # products, IDs and window-local coordinates as the stream below, which one thread parses at ~1 Gbases/s
            size = os.path.getsize(path) if path != "-" and os.path.exists(path) else (1 << 28)
            g = engine.Genome(max(size * (8 if path.endswith(".gz") else 1), 1 << 20), max_records=1 << 16)
            try:
                g.add_fasta(path)
                prods = eng.ScanGenomeChunked(g, cp, sc, chunk, overlap)
                probe_hits = [None] * len(prods)
                if o.probe and prods:
                    # ipcr-probe keeps --chunk-size (internal/probeapp/app.go:108): every product is annotated from its own
                    # amplicon (the rescan reads it from the resident tiles: window-local coordinates are put back by the library)
                    out = (_lib.ProbeHit * len(prods))()
                    _lib.check(_lib.lib().ipcr_probe_products(sc._h, g._h, o.probe.encode(), o.probe_max_mm, out, len(prods)))
                    probe_hits = [out[i] for i in range(len(prods))]
                w, nw = C.POINTER(_lib.ChunkWindow)(), C.c_int64()
                _lib.check(_lib.lib().ipcr_scratch_chunk_windows(sc._h, C.byref(w), C.byref(nw)))
                for p, h in zip(prods, probe_hits):
                    ph = None
                    if h is not None:
                        if o.require_probe and not h.found:                 # internal/visitors/probe.go:20-22
                            continue
                        site = ""
                        if h.found:
                            cw = w[p.Record]
                            amp = g.read(cw.record, cw.start + p.Start, p.End - p.Start)
                            site = amp.upper()[h.pos:h.pos + len(primer.Normalize(o.probe))].decode()
                        ph = (h, site)
                    p = collector.add(path, p)
                    if p is not None:
                        rows.append((path, p, ph))
                g.close()
                continue
            except _lib.IpcrError as e:
                g.close()
                if e.status != _lib.ERR_UNSUPPORTED:                  # (a capped scan that ran in segments: stream the chunks)
                    print(f"error: {e}", file=stderr)
                    continue
        if chunk:
            # the reference's data path: every rolling chunk goes through the engine on its own
            # (ForEachCompiledProduct = ipcr_scan_chunk), the collector restores record coordinates
            from . import fasta
            try:
                for rec in fasta.StreamChunks(path, chunk, overlap):
                    prods = eng.SimulateCompiledWithScratch(rec.ID, rec.Seq, cp, sc)
                    # ipcr-probe keeps --chunk-size (internal/probeapp/app.go:108): the worker that scanned the chunk
                    # annotates its products from the chunk's own tiles, chunk-local coordinates, before the
                    # collector rebases them (pipeline.go:80-89 slices Product.Seq chunk-locally too)
                    hits = sc.probe_products(o.probe, o.probe_max_mm) if o.probe and prods else [None] * len(prods)
                    for p, h in zip(prods, hits):
                        ph = None
                        if h is not None:
                            if o.require_probe and not h.found:             # internal/visitors/probe.go:20-22
                                continue
                            site = ""
                            if h.found:
                                site = bytes(rec.Seq[p.Start:p.End]).upper()[h.pos:h.pos + len(primer.Normalize(o.probe))].decode()
                            ph = (h, site)
                        p = collector.add(path, p)
                        if p is not None:
                            rows.append((path, p, ph))
            except _lib.IpcrError as e:
                print(f"error: {e}", file=stderr)
            continue
        size = os.path.getsize(path) if path != "-" and os.path.exists(path) else (1 << 28)
        factor = 8 if path.endswith(".gz") else 1
        g = engine.Genome(max(size * factor, 1 << 20), max_records=1 << 16)
        try:
            g.add_fasta(path)
        except _lib.IpcrError as e:
            print(f"error: {e}", file=stderr)       # pipeline.go:174-182: record the error, go on
            continue
        prods = eng.ScanGenome(g, cp, sc)
        probe_hits = None
        if o.probe:
            out = (_lib.ProbeHit * max(len(prods), 1))()
            _lib.check(_lib.lib().ipcr_probe_products(sc._h, g._h, o.probe.encode(), o.probe_max_mm, out, len(prods)))
            probe_hits = [out[i] for i in range(len(prods))]
        for i, p in enumerate(prods):
            if probe_hits is None:
                rows.append((path, p, None))
                continue
            h = probe_hits[i]
            if o.require_probe and not h.found:                         # internal/visitors/probe.go:20-22
                continue
            site = ""
            if h.found:
                amp = (g.read(p.Record, p.Start, p.End - p.Start) if p.Start <= p.End else
                       g.read(p.Record, p.Start, g.record_len(p.Record) - p.Start) + g.read(p.Record, 0, p.End))
                site = amp.upper()[h.pos:h.pos + len(primer.Normalize(o.probe))].decode()
            rows.append((path, p, (h, site)))
        g.close()
    if not chunk:  # the collector sees every product in the reference, chunked or not (ids.go quirk included)
        kept = []
        for path, p, ph in rows:
            p = collector.add(path, p)
            if p is not None:
                kept.append((path, p, ph))
        rows = kept
    if o.sort:
        rows.sort(key=lambda t: product_sort_key(t[0], t[1]))
    if o.output == "jsonl" and not o.probe:
        for path, p, _ in rows:
            print(format_jsonl(path, p), file=stdout)
        return o.no_match_exit_code if (not rows and o.no_match_exit_code) else 0
    if not o.no_header:
        print(TSV_HEADER_PROBE if o.probe else TSV_HEADER, file=stdout)
    for path, p, ph in rows:
        line = format_row(path, p)
        if o.probe:                                                      # probeoutput/text.go:11-28
            h, site = ph
            line += "\t" + "\t".join([o.probe_name, o.probe.upper(), "true" if h.found else "false",
                                      chr(h.strand) if h.found else "", str(h.pos) if h.found else "",
                                      str(h.mm) if h.found else "", site])
        print(line, file=stdout)
    if not rows and o.no_match_exit_code:
        return o.no_match_exit_code
    return 0


if __name__ == "__main__":
    sys.exit(run())
