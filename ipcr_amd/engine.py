"""Mirror of the reference's core/engine API for the scan path, over the C ABI.

Same names and argument meaning as the Go package so parity tests read like the
reference's own tests:

    eng = engine.New(engine.Config(MaxMM=1, TerminalWindow=3))
    products = eng.SimulateBatch("seq", b"ACGT...", [primer.Pair(...)])

Everything here is thin glue; panel compilation, the scan, hit ordering/capping and the
amplicon join all live behind libipcr_hip.so (ipcr_amd/csrc).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

from . import _lib
from .primer import Oligo, Pair, SelfPairs


@dataclass
class Config:
    """engine.Config -- core/engine/engine.go:10-19"""
    MaxMM: int = 0
    TerminalWindow: int = 0
    MinLen: int = 0
    MaxLen: int = 0
    HitCap: int = 0
    NeedSites: bool = False
    SeedLen: int = 0
    Circular: bool = False

    def _c(self) -> _lib.Config:
        return _lib.Config(self.MaxMM, self.TerminalWindow, self.MinLen, self.MaxLen, self.HitCap,
                           self.SeedLen, 1 if self.Circular else 0, 1 if self.NeedSites else 0)


@dataclass
class Product:
    """engine.Product -- core/engine/product.go:4-35 (scan-produced fields)"""
    ExperimentID: str
    SequenceID: str
    Start: int
    End: int
    Length: int
    Type: str
    FwdMM: int
    RevMM: int
    FwdMismatchIdx: Tuple[int, ...]
    RevMismatchIdx: Tuple[int, ...]
    FwdPrimer: str = ""
    RevPrimer: str = ""
    FwdSite: str = ""
    RevSite: str = ""
    Record: int = 0

    def sig(self):
        """core/engine/approx_seed_oracle_test.go:12-41 signature (minus SequenceID)."""
        return (self.ExperimentID, self.Start, self.End, self.Length, self.Type, self.FwdMM,
                self.RevMM, self.FwdMismatchIdx, self.RevMismatchIdx)


@dataclass
class Hit:
    """primer.Match as found on the device, per distinct pattern."""
    Record: int
    Pattern: int
    Pos: int
    Mismatches: int
    MismatchIdx: Tuple[int, ...]
    SeedSpanReset: bool


def _idx_from_mask(m0: int, m1: int) -> Tuple[int, ...]:
    out = []
    for w, m in enumerate((m0, m1)):
        while m:
            b = (m & -m).bit_length() - 1
            out.append(b + 64 * w)
            m &= m - 1
    return tuple(out)


class CompiledPanel:
    """engine.CompiledPanel -- core/engine/compiled.go:77-91 (exported fields Pairs, Cfg)."""

    def __init__(self, cfg: Config, pairs: Sequence[Pair]):
        self.Cfg = cfg
        self.Pairs = list(pairs)
        n = len(self.Pairs)
        self._keep = [(p.ID.encode(), p.Forward.encode(), p.Reverse.encode()) for p in self.Pairs]
        arr = (_lib.Pair * max(n, 1))()
        for i, p in enumerate(self.Pairs):
            arr[i] = _lib.Pair(self._keep[i][0], self._keep[i][1], self._keep[i][2], p.MinProduct, p.MaxProduct)
        h = C.c_void_p()
        cc = cfg._c()
        _lib.check(_lib.lib().ipcr_panel_create(C.byref(cc), arr, n, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().ipcr_panel_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def have(self, pair: int, which: str) -> bool:
        """compiledHas(cp.Have, pair, which) -- core/engine/compiled.go:35-37"""
        return bool(_lib.lib().ipcr_panel_have(self._h, pair, which.encode()))

    @property
    def num_patterns(self) -> int:
        return _lib.lib().ipcr_panel_num_patterns(self._h)

    @property
    def num_patterns_total(self) -> int:
        return _lib.lib().ipcr_panel_num_patterns_total(self._h)

    def pattern_info(self, pattern: int):
        """(sequence, window_on_left, window_bases_enforced_on_device, seed_off, seed_len)."""
        buf = C.create_string_buffer(_lib.IPCR_MAX_PRIMER_LEN + 1)
        left, tw, so, sl = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().ipcr_panel_pattern_info(self._h, pattern, buf, len(buf), C.byref(left), C.byref(tw),
                                                      C.byref(so), C.byref(sl)))
        return buf.value.decode(), bool(left.value), tw.value, so.value, sl.value

    def slot_pattern(self, pair: int, which: str, mode: int = 0) -> int:
        return _lib.lib().ipcr_panel_slot_pattern(self._h, pair, which.encode(), mode)

    def filter_source(self, mode: int = 0) -> str:
        """HIP source of the panel-specialised filter ('' when the panel is not specialisable)."""
        need = C.c_size_t()
        _lib.check(_lib.lib().ipcr_panel_filter_source(self._h, mode, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _lib.check(_lib.lib().ipcr_panel_filter_source(self._h, mode, buf, need.value, None))
        return buf.value.decode()

    def set_shard(self, index: int, count: int) -> None:
        """Pattern-axis sharding: this panel object scans every count-th distinct pattern only (before the first scan);
        hits keep panel-wide pattern ids, so gathered hit lists join to the unsharded result (dist.py)."""
        _lib.check(_lib.lib().ipcr_panel_set_shard(self._h, index, count))

    def scanned_patterns(self, mode: int = 0) -> List[int]:
        n = _lib.lib().ipcr_panel_scanned_patterns(self._h, mode, None, 0)
        out = (C.c_int32 * max(n, 1))()
        _lib.lib().ipcr_panel_scanned_patterns(self._h, mode, out, n)
        return [out[i] for i in range(n)]

    def wait_ready(self) -> None:
        """block until the kernels of a small panel, built in the background, are in use (ipcr_panel_wait_ready)"""
        _lib.check(_lib.lib().ipcr_panel_wait_ready(self._h))

    @property
    def device_slots(self) -> int:
        """devices this panel holds tables and kernels on"""
        return int(_lib.lib().ipcr_panel_device_slots(self._h))

    def set_specialize(self, enable: bool) -> None:
        _lib.check(_lib.lib().ipcr_panel_set_specialize(self._h, 1 if enable else 0))


class SimulationScratch:
    """engine.SimulationScratch -- core/engine/hit_collect.go:12-34: one HIP stream + device
    staging buffers per worker; never shared between workers."""

    def __init__(self, cp: CompiledPanel, host_only: bool = False, device: Optional[int] = None):
        """device: the GPU this worker's stream and buffers live on (ipcr_scratch_create_on); None = the process default
        (ipcr_set_device, else the calling thread's current HIP device)"""
        self._cp = cp
        h = C.c_void_p()
        if host_only:  # results of JoinHits only; cannot scan
            _lib.check(_lib.lib().ipcr_scratch_create_host(cp._h, C.byref(h)))
        elif device is None:
            _lib.check(_lib.lib().ipcr_scratch_create(cp._h, C.byref(h)))
        else:
            _lib.check(_lib.lib().ipcr_scratch_create_on(cp._h, int(device), C.byref(h)))
        self._h = h

    @property
    def device(self) -> int:
        return int(_lib.lib().ipcr_scratch_device(self._h))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().ipcr_scratch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def chain_after(self, prev: "SimulationScratch") -> None:
        """The next scan begun on this scratch sweeps the tiles only after `prev`'s current sweep."""
        _lib.check(_lib.lib().ipcr_scratch_chain_after(self._h, prev._h))

    def stats(self) -> _lib.ScanStats:
        st = _lib.ScanStats()
        _lib.check(_lib.lib().ipcr_scratch_stats(self._h, C.byref(st)))
        return st

    def hits(self) -> List[Hit]:
        ptr = C.POINTER(_lib.Hit)()
        n = C.c_int64()
        _lib.check(_lib.lib().ipcr_scratch_hits(self._h, C.byref(ptr), C.byref(n)))
        out = []
        for i in range(n.value):
            h = ptr[i]
            idx = _idx_from_mask(h.mm_mask[0], h.mm_mask[1])
            out.append(Hit(h.record, h.pattern & 0x7FFFFFFF, h.pos, len(idx), idx, bool(h.pattern >> 31)))
        return out

    def raw_hits(self):
        """(pointer, count) of the ipcr_hit array of the last scan."""
        ptr = C.POINTER(_lib.Hit)()
        n = C.c_int64()
        _lib.check(_lib.lib().ipcr_scratch_hits(self._h, C.byref(ptr), C.byref(n)))
        return ptr, n.value

    def device_hits(self):
        """(device address of the 64-byte header + hit slots, hits of the last scan, capacity in hits):
        ipcr_scratch_device_hits, for a device-to-device exchange."""
        ptr = C.c_void_p()
        n = C.c_uint64()
        cap = C.c_uint64()
        _lib.check(_lib.lib().ipcr_scratch_device_hits(self._h, C.byref(ptr), C.byref(n), C.byref(cap)))
        return ptr.value, n.value, cap.value

    def num_products(self) -> int:
        ptr = C.POINTER(_lib.Product)()
        n = C.c_int64()
        _lib.check(_lib.lib().ipcr_scratch_products(self._h, C.byref(ptr), C.byref(n)))
        return n.value

    def products(self, seq_ids: Sequence[str]) -> List[Product]:
        ptr = C.POINTER(_lib.Product)()
        n = C.c_int64()
        _lib.check(_lib.lib().ipcr_scratch_products(self._h, C.byref(ptr), C.byref(n)))
        return [_product(self._cp, ptr[i], seq_ids) for i in range(n.value)]

    def probe_products(self, probe: str, max_mm: int):
        """ipcr_probe_scratch_products: oligo.BestHit (core/oligo/oligo.go:19-77) for every product of the last
        ipcr_scan_chunk on this scratch, rescanned from the chunk's own tiles -- what visitors.Probe.Visit computes
        from Product.Seq (internal/visitors/probe.go:18-33).  List of _lib.ProbeHit, one per product."""
        n = self.num_products()
        out = (_lib.ProbeHit * max(n, 1))()
        _lib.check(_lib.lib().ipcr_probe_scratch_products(self._h, probe.encode(), max_mm, out, n))
        return [out[i] for i in range(n)]


def _fill_sites(pr: Product, seq: bytes) -> None:
    """FwdSite / RevSite as core/engine/engine.go:175-183 slices them (NeedSites, pretty text only):
    the target under the left primer, and the reverse complement of the target under the right one."""
    from .primer import RevComp
    flen, rlen = len(pr.FwdPrimer), len(pr.RevPrimer)
    if pr.Start + flen <= len(seq):
        pr.FwdSite = seq[pr.Start:pr.Start + flen].decode("latin-1")
    b = pr.End - rlen
    if 0 <= b and pr.End <= len(seq):
        pr.RevSite = RevComp(seq[b:pr.End]).decode("latin-1")   # raises where the reference panics (rc.go:27-34)


def _product(cp: CompiledPanel, p: _lib.Product, seq_ids: Sequence[str]) -> Product:
    pair = cp.Pairs[p.pair]
    fwd = p.type == 0
    return Product(
        ExperimentID=pair.ID,
        SequenceID=seq_ids[p.record] if p.record < len(seq_ids) else str(p.record),
        Start=p.start, End=p.end, Length=p.length,
        Type="forward" if fwd else "revcomp",
        FwdMM=p.fwd_mm, RevMM=p.rev_mm,
        FwdMismatchIdx=tuple(p.fwd_idx[k] for k in range(p.n_fwd_idx)),
        RevMismatchIdx=tuple(p.rev_idx[k] for k in range(p.n_rev_idx)),
        FwdPrimer=pair.Forward if fwd else pair.Reverse,   # core/engine/engine.go:197-198,385-386
        RevPrimer=pair.Reverse if fwd else pair.Forward,
        Record=p.record,
    )


class Genome:
    """Reference records packed once into 2-bit + invalid-bit tiles resident in HBM."""

    def __init__(self, capacity_bases: int, max_records: int = 1, device: Optional[int] = None):
        h = C.c_void_p()
        if device is None:
            _lib.check(_lib.lib().ipcr_genome_create(capacity_bases, max_records, C.byref(h)))
        else:
            _lib.check(_lib.lib().ipcr_genome_create_on(capacity_bases, max_records, int(device), C.byref(h)))
        self._h = h
        self.ids: List[str] = []

    @property
    def device(self) -> int:
        return int(_lib.lib().ipcr_genome_device(self._h))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().ipcr_genome_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_record(self, seq_id: str, seq) -> None:
        b = seq if isinstance(seq, (bytes, bytearray)) else seq.encode()
        _lib.check(_lib.lib().ipcr_genome_add_record(self._h, bytes(b), len(b)))
        self.ids.append(seq_id)

    def add_record_device(self, seq_id: str, dev_ptr: int, length: int) -> None:
        _lib.check(_lib.lib().ipcr_genome_add_record_device(self._h, C.c_void_p(dev_ptr), length))
        self.ids.append(seq_id)

    def add_fasta(self, path: str) -> int:
        """Pack every record of a FASTA file (plain / gzip); the device normalises the raw text
        (core/fasta/scan.go:10-69, normalize.go:5-14).  Returns the number of records added."""
        n = C.c_uint32()
        first = self.num_records
        _lib.check(_lib.lib().ipcr_genome_add_fasta(self._h, path.encode(), C.byref(n), None, 0, None))
        for r in range(first, first + n.value):
            self.ids.append(_lib.lib().ipcr_genome_record_id(self._h, r).decode())
        return n.value

    def read(self, record: int, pos: int, length: int) -> bytes:
        buf = C.create_string_buffer(max(length, 1))
        _lib.check(_lib.lib().ipcr_genome_read(self._h, record, pos, buf, length))
        return buf.raw[:length]

    @property
    def num_records(self) -> int:
        return _lib.lib().ipcr_genome_num_records(self._h)

    def record_len(self, r: int) -> int:
        return _lib.lib().ipcr_genome_record_len(self._h, r)

    def record_flags(self, r: int) -> int:
        return _lib.lib().ipcr_genome_record_flags(self._h, r)

    @property
    def total_bases(self) -> int:
        return _lib.lib().ipcr_genome_total_bases(self._h)

    @property
    def tile_bytes(self) -> int:
        return _lib.lib().ipcr_genome_tile_bytes(self._h)

    @property
    def pack_ms(self) -> float:
        return _lib.lib().ipcr_genome_pack_ms(self._h)


def lcg_fill_device(dev_ptr: int, length: int, seed: int, stream_offset: int = 0) -> None:
    """benchDNA on the device -- core/engine/performance_benchmark_test.go:67-76: bases
    [stream_offset, stream_offset + length) of the LCG stream started from `seed`."""
    _lib.check(_lib.lib().ipcr_lcg_fill_device(C.c_void_p(dev_ptr), length, seed & 0xFFFFFFFF, stream_offset))


class Engine:
    """engine.Engine -- core/engine/engine.go:21-30"""

    def __init__(self, cfg: Config):
        self.cfg = cfg

    def SetHitCap(self, n: int) -> None:  # engine.go:30
        self.cfg = Config(**{**self.cfg.__dict__, "HitCap": n})

    # -- compiled.go:96-136
    def CompilePanel(self, pairs: Sequence[Pair]) -> CompiledPanel:
        return CompiledPanel(self.cfg, pairs)

    # -- hit_collect.go:31-34
    def NewSimulationScratch(self, cp: CompiledPanel, device: Optional[int] = None) -> SimulationScratch:
        return SimulationScratch(cp, device=device)

    # -- compiled.go:162-267
    def ForEachCompiledProduct(self, seqID: str, seq, cp: Optional[CompiledPanel],
                               scratch: Optional[SimulationScratch],
                               emit: Optional[Callable[[Product], Optional[Exception]]]):
        """Calls emit(product) in the reference's emission order.  A truthy return from emit
        aborts the scan and is returned (the reference returns emit's error)."""
        if cp is None or not cp.Pairs or emit is None:  # compiled.go:163-165
            return None
        own = scratch is None
        if own:
            scratch = SimulationScratch(cp)
        b = seq if isinstance(seq, (bytes, bytearray)) else seq.encode()
        err_box = []

        def _cb(pp, _user):
            pr = _product(cp, pp.contents, [seqID])
            if cp.Cfg.NeedSites:
                _fill_sites(pr, bytes(b))
            r = emit(pr)
            if r:
                err_box.append(r)
                return 1
            return 0

        cb = _lib.EMIT_FN(_cb)
        st = _lib.lib().ipcr_scan_chunk(cp._h, scratch._h, bytes(b), len(b), C.cast(cb, C.c_void_p), None)
        if own:
            scratch.close()
        if st == _lib.ERR_ABORTED and err_box:
            return err_box[0]
        _lib.check(st)
        return None

    # -- compiled.go:148-156
    def SimulateCompiledWithScratch(self, seqID: str, seq, cp: CompiledPanel,
                                    scratch: Optional[SimulationScratch]) -> List[Product]:
        if cp is None or not cp.Pairs:
            return []
        own = scratch is None
        if own:
            scratch = SimulationScratch(cp)
        b = seq if isinstance(seq, (bytes, bytearray)) else seq.encode()
        _lib.check(_lib.lib().ipcr_scan_chunk(cp._h, scratch._h, bytes(b), len(b), None, None))
        out = scratch.products([seqID])
        if cp.Cfg.NeedSites:
            for pr in out:
                _fill_sites(pr, bytes(b))
        if own:
            scratch.close()
        return out

    # -- compiled.go:141-143
    def SimulateCompiled(self, seqID: str, seq, cp: CompiledPanel) -> List[Product]:
        return self.SimulateCompiledWithScratch(seqID, seq, cp, None)

    # -- engine.go:49-51
    def SimulateBatch(self, seqID: str, seq, pairs: Sequence[Pair]) -> List[Product]:
        cp = self.CompilePanel(pairs)
        try:
            return self.SimulateCompiled(seqID, seq, cp)
        finally:
            cp.close()

    # -- engine.go:33-36
    def Simulate(self, seqID: str, seq, p: Pair) -> List[Product]:
        return self.SimulateBatch(seqID, seq, [p])

    # -- self.go:10-16
    def SimulateSelf(self, seqID: str, seq, oligos: Sequence[Oligo]) -> List[Product]:
        if not oligos:
            return []
        return self.SimulateBatch(seqID, seq, SelfPairs(oligos))

    # -- resident-genome form of ForEachCompiledProduct: all records in one launch
    def ScanGenome(self, genome: Genome, cp: CompiledPanel, scratch: SimulationScratch) -> List[Product]:
        _lib.check(_lib.lib().ipcr_scan_genome(cp._h, scratch._h, genome._h, None, None))
        return scratch.products(genome.ids)

    def ScanGenomeChunked(self, genome: Genome, cp: CompiledPanel, scratch: SimulationScratch, chunkSize: int,
                          overlap: int) -> List[Product]:
        """The resident genome scanned as the pipeline scans it under --chunk-size (core/fasta/path_ctx.go:83-179 windows,
        one Engine.ForEachCompiledProduct call per window: internal/pipeline/pipeline.go:60-125) -- from ONE sweep of the
        tiles.  Products carry window-local coordinates and the window's ID ('id:start-end', or the record's own ID when
        it never filled a window), exactly what fasta.StreamChunks + SimulateCompiledWithScratch give chunk by chunk."""
        _lib.check(_lib.lib().ipcr_scan_genome_chunked(cp._h, scratch._h, genome._h, chunkSize, overlap, None, None))
        w, n = C.POINTER(_lib.ChunkWindow)(), C.c_int64()
        _lib.check(_lib.lib().ipcr_scratch_chunk_windows(scratch._h, C.byref(w), C.byref(n)))
        ids = genome.ids
        names = [ids[w[i].record] if w[i].plain else "%s:%d-%d" % (ids[w[i].record], w[i].start, w[i].end) for i in range(n.value)]
        return scratch.products(names)

    def ScanGenomeCount(self, genome: Genome, cp: CompiledPanel, scratch: SimulationScratch) -> int:
        """Same scan + join, products left in the scratch (no Python object per product)."""
        _lib.check(_lib.lib().ipcr_scan_genome(cp._h, scratch._h, genome._h, None, None))
        return scratch.num_products()

    def JoinHits(self, cp: CompiledPanel, scratch: SimulationScratch, hits, record_len: Sequence[int],
                 record_flags: Sequence[int], seq_ids: Optional[Sequence[str]] = None) -> List[Product]:
        """Join step alone (core/engine/engine.go:108-404) over hit records, e.g. the all-gathered
        hits of several GPUs.  `hits` is a numpy array of dist.HIT_DTYPE (or anything exposing
        ctypes.data / len)."""
        n = len(hits)
        nrec = len(record_len)
        lens = (C.c_uint64 * max(nrec, 1))(*record_len)
        flags = (C.c_uint8 * max(nrec, 1))(*record_flags)
        ptr = C.c_void_p(hits.ctypes.data) if n else None
        _lib.check(_lib.lib().ipcr_join_hits(cp._h, scratch._h, ptr, n, lens, flags, nrec, None, None))
        ids = list(seq_ids) if seq_ids is not None else [str(r) for r in range(nrec)]
        return scratch.products(ids)

    def ScanGenomeBegin(self, genome: Genome, cp: CompiledPanel, scratch: SimulationScratch) -> None:
        """Enqueue a scan of the resident genome on `scratch` and return at once (pipelining)."""
        _lib.check(_lib.lib().ipcr_scan_genome_begin(cp._h, scratch._h, genome._h))

    def ScanGenomeEndCount(self, genome: Genome, cp: CompiledPanel, scratch: SimulationScratch) -> int:
        """Wait for the scan begun on `scratch`, join; products stay in the scratch."""
        _lib.check(_lib.lib().ipcr_scan_genome_end(cp._h, scratch._h, genome._h, None, None))
        return scratch.num_products()

    def ScanGenomeHits(self, genome: Genome, cp: CompiledPanel, scratch: SimulationScratch) -> List[Hit]:
        _lib.check(_lib.lib().ipcr_scan_genome_hits(cp._h, scratch._h, genome._h))
        return scratch.hits()


def New(c: Config) -> Engine:
    """engine.New -- core/engine/engine.go:27"""
    return Engine(c)
