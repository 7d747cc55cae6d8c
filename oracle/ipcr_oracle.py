"""ctypes binding of the CPU oracle (oracle/libipcr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package (ipcr_amd/) must never import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libipcr_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ipcr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("max_mm", "terminal_window", "min_len", "max_len", "hit_cap", "seed_len", "circular")]


class _Match(C.Structure):
    _fields_ = [("pos", C.c_int32), ("mm", C.c_int32), ("len", C.c_int32), ("nidx", C.c_int32),
                ("idx", C.POINTER(C.c_int32))]


class _Matches(C.Structure):
    _fields_ = [("v", C.POINTER(_Match)), ("n", C.c_int32), ("cap", C.c_int32)]


class _Product(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("pair", "start", "end", "length", "type", "fwd_mm", "rev_mm", "nf", "nr")] + \
               [("fidx", C.POINTER(C.c_int32)), ("ridx", C.POINTER(C.c_int32))]


class _Products(C.Structure):
    _fields_ = [("v", C.POINTER(_Product)), ("n", C.c_int32), ("cap", C.c_int32)]


class _Hit(C.Structure):
    _fields_ = [("found", C.c_int32), ("strand", C.c_int32), ("pos", C.c_int32), ("mm", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.or_iupac_mask.restype = C.c_uint8
        L.or_iupac_mask.argtypes = [C.c_uint8]
        L.or_base_match.argtypes = [C.c_uint8, C.c_uint8]
        L.or_revcomp.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.or_mismatch_count.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.or_find_matches.restype = None
        L.or_find_matches.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(_Matches)]
        L.or_matches_free.argtypes = [C.POINTER(_Matches)]
        L.or_sort_and_bounds.restype = None
        L.or_sort_and_bounds.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.or_validate_primer.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.or_products_free.argtypes = [C.POINTER(_Products)]
        L.or_simulate_bruteforce.restype = None
        L.or_simulate_bruteforce.argtypes = [C.POINTER(_Config), C.c_char_p, C.c_int, C.c_int,
                                             C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(_Products)]
        L.or_panel_create.restype = C.c_void_p
        L.or_panel_create.argtypes = [C.POINTER(_Config), C.c_int, C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.or_panel_free.argtypes = [C.c_void_p]
        L.or_panel_have.argtypes = [C.c_void_p, C.c_int, C.c_char]
        L.or_panel_num_seed_patterns.argtypes = [C.c_void_p]
        L.or_panel_num_nodes.argtypes = [C.c_void_p]
        L.or_panel_seed_pattern.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.or_panel_scan.restype = None
        L.or_panel_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Products)]
        L.or_panel_scan_matches.restype = None
        L.or_panel_scan_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_char,
                                            C.POINTER(_Matches)]
        L.or_ac_scan.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int,
                                 C.POINTER(C.c_int32), C.c_int]
        L.or_non_acgt_ranges.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.c_int]
        L.or_halo_starts.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.c_int]
        L.or_build_seed_patterns_count.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                                   C.c_int, C.c_int, C.c_int]
        L.or_best_hit.restype = _Hit
        L.or_best_hit.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.or_bench_dna.restype = None
        L.or_bench_dna.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
        L.or_bench_primer.restype = None
        L.or_bench_primer.argtypes = [C.c_int, C.c_int, C.c_char_p]
        L.or_different_base.restype = C.c_uint8
        L.or_different_base.argtypes = [C.c_uint8]
        L.or_make_bench_fixture.restype = C.c_int64
        L.or_make_bench_fixture.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                            C.c_char_p, C.c_char_p]
        L.or_baseline_scan_mt.restype = C.c_int64
        L.or_baseline_scan_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int]
        L.or_baseline_scan_pool.restype = C.c_int64
        L.or_baseline_scan_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = L
    return _lib


@dataclass
class Config:
    """engine.Config (core/engine/engine.go:10-19)."""
    max_mm: int = 0
    terminal_window: int = 0
    min_len: int = 0
    max_len: int = 0
    hit_cap: int = 0
    seed_len: int = 0
    circular: bool = False

    def c(self) -> _Config:
        return _Config(self.max_mm, self.terminal_window, self.min_len, self.max_len,
                       self.hit_cap, self.seed_len, 1 if self.circular else 0)


@dataclass
class Pair:
    """primer.Pair (core/primer/pair.go:4-10)."""
    id: str
    forward: str
    reverse: str
    min_product: int = 0
    max_product: int = 0


@dataclass
class Match:
    pos: int
    mm: int
    length: int
    idx: Tuple[int, ...]


@dataclass
class Product:
    experiment_id: str
    start: int
    end: int
    length: int
    type: str
    fwd_mm: int
    rev_mm: int
    fwd_idx: Tuple[int, ...]
    rev_idx: Tuple[int, ...]

    def sig(self):
        """Signature tuple of core/engine/approx_seed_oracle_test.go:12-41 (minus SequenceID)."""
        return (self.experiment_id, self.start, self.end, self.length, self.type, self.fwd_mm,
                self.rev_mm, self.fwd_idx, self.rev_idx)


def _b(s) -> bytes:
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


def _pairs_c(pairs: Sequence[Pair]):
    n = len(pairs)
    fwd = (C.c_char_p * max(n, 1))(*[_b(p.forward) for p in pairs])
    rev = (C.c_char_p * max(n, 1))(*[_b(p.reverse) for p in pairs])
    mn = (C.c_int32 * max(n, 1))(*[p.min_product for p in pairs])
    mx = (C.c_int32 * max(n, 1))(*[p.max_product for p in pairs])
    return n, fwd, rev, mn, mx


def _matches_out(ms: _Matches) -> List[Match]:
    out = []
    for i in range(ms.n):
        m = ms.v[i]
        out.append(Match(m.pos, m.mm, m.len, tuple(m.idx[k] for k in range(m.nidx))))
    lib().or_matches_free(C.byref(ms))
    return out


def _products_out(ps: _Products, pairs: Sequence[Pair]) -> List[Product]:
    out = []
    for i in range(ps.n):
        p = ps.v[i]
        out.append(Product(pairs[p.pair].id, p.start, p.end, p.length,
                           "forward" if p.type == 0 else "revcomp", p.fwd_mm, p.rev_mm,
                           tuple(p.fidx[k] for k in range(p.nf)), tuple(p.ridx[k] for k in range(p.nr))))
    lib().or_products_free(C.byref(ps))
    return out


def iupac_mask(c: str) -> int:
    return lib().or_iupac_mask(ord(c))


def base_match(g: str, p: str) -> bool:
    return bool(lib().or_base_match(ord(g), ord(p)))


def revcomp(s) -> bytes:
    s = _b(s)
    out = C.create_string_buffer(len(s) + 1)
    bad = lib().or_revcomp(s, len(s), out)
    if bad:
        raise ValueError(f"invalid reverse-complement base at position {bad}")
    return out.raw[:len(s)]


def mismatch_count(g, p) -> int:
    g, p = _b(g), _b(p)
    if len(g) != len(p):
        raise ValueError("MismatchCount: length mismatch")
    return lib().or_mismatch_count(g, p, len(p))


def find_matches(seq, primer, max_mm: int, cap_hits: int, tw: int) -> List[Match]:
    seq, primer = _b(seq), _b(primer)
    ms = _Matches()
    lib().or_find_matches(seq, len(seq), primer, len(primer), max_mm, cap_hits, tw, C.byref(ms))
    return _matches_out(ms)


def sort_and_bounds(positions: Sequence[int], query: int):
    """sortMatchesByPos + lowerBoundMatchPos / upperBoundMatchPos -- core/engine/engine.go:70-93.
    -> (sorted positions, their input indices (stability), lo, hi)"""
    n = len(positions)
    arr = (C.c_int32 * max(n, 1))(*positions)
    out = (C.c_int32 * max(2 * n, 1))()
    lo, hi = C.c_int(), C.c_int()
    lib().or_sort_and_bounds(arr, n, query, out, C.byref(lo), C.byref(hi))
    return [out[2 * i] for i in range(n)], [out[2 * i + 1] for i in range(n)], lo.value, hi.value


def validate_primer(raw: str) -> str:
    """primer.Validate -- core/primer/validate.go:27-37; ValueError where the reference returns an error."""
    buf = C.create_string_buffer(len(raw.encode()) + 2)
    r = lib().or_validate_primer(raw.encode(), buf, len(buf))
    if r == 0:
        raise ValueError("empty primer")
    if r < 0:
        raise ValueError("invalid primer base at position %d" % -r)
    return buf.value.decode()


def simulate_bruteforce(cfg: Config, seq, pairs: Sequence[Pair]) -> List[Product]:
    seq = _b(seq)
    n, fwd, rev, mn, mx = _pairs_c(pairs)
    ps = _Products()
    cc = cfg.c()
    lib().or_simulate_bruteforce(C.byref(cc), seq, len(seq), n, fwd, rev, mn, mx, C.byref(ps))
    return _products_out(ps, pairs)


class Panel:
    """engine.CompiledPanel + Engine.SimulateCompiled (core/engine/compiled.go)."""

    def __init__(self, cfg: Config, pairs: Sequence[Pair]):
        self.cfg = cfg
        self.pairs = list(pairs)
        n, fwd, rev, mn, mx = _pairs_c(self.pairs)
        cc = cfg.c()
        self._h = lib().or_panel_create(C.byref(cc), n, fwd, rev, mn, mx)

    def close(self):
        if self._h:
            lib().or_panel_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def have(self, pair: int, which: str) -> bool:
        return bool(lib().or_panel_have(self._h, pair, which.encode()))

    @property
    def num_seed_patterns(self) -> int:
        return lib().or_panel_num_seed_patterns(self._h)

    @property
    def num_nodes(self) -> int:
        return lib().or_panel_num_nodes(self._h)

    def seed_patterns(self) -> List[Tuple[str, int]]:
        out = []
        buf = C.create_string_buffer(64)
        npay = C.c_int()
        for i in range(self.num_seed_patterns):
            lib().or_panel_seed_pattern(self._h, i, buf, 64, C.byref(npay))
            out.append((buf.value.decode(), npay.value))
        return out

    def scan(self, seq) -> List[Product]:
        seq = _b(seq)
        ps = _Products()
        lib().or_panel_scan(self._h, seq, len(seq), C.byref(ps))
        return _products_out(ps, self.pairs)

    def scan_ptr(self, ptr: int, n: int) -> List[Product]:
        ps = _Products()
        lib().or_panel_scan(self._h, C.c_void_p(ptr), n, C.byref(ps))
        return _products_out(ps, self.pairs)

    def scan_matches(self, seq, pair: int, which: str) -> List[Match]:
        seq = _b(seq)
        ms = _Matches()
        lib().or_panel_scan_matches(self._h, seq, len(seq), pair, which.encode(), C.byref(ms))
        return _matches_out(ms)

    def baseline_scan_pool(self, ptr: int, n: int, chunk_size: int, overlap: int, threads: int, passes: int):
        """`passes` passes over one record served by one pool of worker threads -> (distinct products of a pass,
        workers that scanned at least one chunk, chunks per pass)"""
        busy, nch = C.c_int(), C.c_int()
        uniq = lib().or_baseline_scan_pool(self._h, C.c_void_p(ptr), n, chunk_size, overlap, threads, passes,
                                           C.byref(busy), C.byref(nch))
        return int(uniq), busy.value, nch.value

    def baseline_scan_mt(self, ptr: int, n: int, chunk_size: int, overlap: int, threads: int) -> int:
        return lib().or_baseline_scan_mt(self._h, C.c_void_p(ptr), n, chunk_size, overlap, threads)


def simulate_batch(cfg: Config, seq, pairs: Sequence[Pair]) -> List[Product]:
    """Engine.SimulateBatch (core/engine/engine.go:49-51)."""
    p = Panel(cfg, pairs)
    try:
        return p.scan(seq)
    finally:
        p.close()


def ac_scan(patterns: Sequence[str], seq) -> List[Tuple[int, int]]:
    seq = _b(seq)
    arr = (C.c_char_p * len(patterns))(*[_b(p) for p in patterns])
    cap = 4096
    out = (C.c_int32 * (2 * cap))()
    n = lib().or_ac_scan(len(patterns), arr, seq, len(seq), out, cap)
    return [(out[2 * i], out[2 * i + 1]) for i in range(min(n, cap))]


def non_acgt_ranges(seq) -> List[Tuple[int, int]]:
    seq = _b(seq)
    cap = 4096
    out = (C.c_int32 * (2 * cap))()
    n = lib().or_non_acgt_ranges(seq, len(seq), out, cap)
    return [(out[2 * i], out[2 * i + 1]) for i in range(min(n, cap))]


def halo_starts(seq_len: int, primer_len: int, ranges: Sequence[Tuple[int, int]]) -> List[int]:
    flat = (C.c_int32 * max(2 * len(ranges), 1))(*[v for r in ranges for v in r])
    cap = 65536
    out = (C.c_int32 * cap)()
    n = lib().or_halo_starts(seq_len, primer_len, len(ranges), flat, out, cap)
    return [out[i] for i in range(min(n, cap))]


def build_seed_patterns_count(pairs: Sequence[Pair], seed_len: int, tw: int, max_mm: int) -> int:
    n, fwd, rev, _, _ = _pairs_c(pairs)
    return lib().or_build_seed_patterns_count(n, fwd, rev, seed_len, tw, max_mm)


@dataclass
class Hit:
    found: bool = False
    strand: str = ""
    pos: int = 0
    mm: int = 0
    site: str = ""


def best_hit(amplicon, probe: str, max_mm: int) -> Hit:
    """oligo.BestHit (core/oligo/oligo.go:19-77)."""
    amp = _b(amplicon)
    h = lib().or_best_hit(amp, len(amp), _b(probe), max_mm)
    if not h.found:
        return Hit()
    plen = len("".join(ch for ch in probe if not ch.isspace() and ch not in "'\""))
    site = amp.upper()[h.pos:h.pos + plen].decode() if h.pos + plen <= len(amp) else ""
    return Hit(True, chr(h.strand), h.pos, h.mm, site)


def bench_dna(n: int, seed: int) -> bytes:
    buf = C.create_string_buffer(n)
    lib().or_bench_dna(buf, n, seed)
    return buf.raw[:n]


def bench_dna_into(ptr: int, n: int, seed: int) -> None:
    lib().or_bench_dna(C.c_void_p(ptr), n, seed)


def bench_primer(idx: int, n: int = 20) -> str:
    buf = C.create_string_buffer(n + 1)
    lib().or_bench_primer(idx, n, buf)
    return buf.value.decode()


def different_base(b: str) -> str:
    return chr(lib().or_different_base(ord(b)))


def make_bench_fixture(pair_count: int, genome_len: int, mutate_forward: bool, reference_n: bool):
    """makeEngineBenchFixture (core/engine/performance_benchmark_test.go:25-65)."""
    pc = max(pair_count, 1)
    glen = max(genome_len, 256 + pc * 256 + 180)
    seq = C.create_string_buffer(glen)
    fwd = C.create_string_buffer(pc * 21)
    rev = C.create_string_buffer(pc * 21)
    n = lib().or_make_bench_fixture(pair_count, genome_len, int(mutate_forward), int(reference_n),
                                    seq, fwd, rev)
    pairs = []
    for i in range(pc):
        pairs.append(Pair("bench_%03d" % i, fwd.raw[i * 21:i * 21 + 20].decode(),
                          rev.raw[i * 21:i * 21 + 20].decode(), 128, 180 + 32))
    return seq.raw[:n], pairs
