"""Mirror of internal/visitors/nested.go for the scan path: the inner PCR of every outer amplicon, batched.

The reference builds a fresh engine and scans one amplicon per call on the collector goroutine
(nested.go:17-21); here the amplicons of all outer products are gathered on the device, packed as records of a
scratch-private genome and scanned by the compiled inner panel in one launch (ipcr_nested_windows)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from . import _lib, engine


@dataclass
class NestedProduct:
    """nestedoutput.NestedProduct -- the outer product plus its best inner product (amplicon coordinates)"""
    Product: Optional[engine.Product]
    InnerFound: bool
    InnerPairID: str = ""
    InnerStart: int = 0
    InnerEnd: int = 0
    InnerLength: int = 0
    InnerType: str = ""
    InnerFwdMM: int = 0
    InnerRevMM: int = 0


def _convert(out, n: int, cp: engine.CompiledPanel, products: Optional[Sequence[engine.Product]]) -> List[NestedProduct]:
    res = []
    for i in range(n):
        h = out[i]
        p = products[i] if products is not None else None
        if not h.found:
            res.append(NestedProduct(p, False))
            continue
        res.append(NestedProduct(p, True, cp.Pairs[h.pair].ID, h.start, h.end, h.length,
                                 "forward" if h.type == 0 else "revcomp", h.fwd_mm, h.rev_mm))
    return res


def NestedWindows(genome: engine.Genome, windows: Sequence[Tuple[int, int, int]], inner: engine.CompiledPanel,
                  inner_scratch: engine.SimulationScratch) -> List[NestedProduct]:
    """windows = (record, start, end) amplicons of the resident genome; start > end spans the origin"""
    n = len(windows)
    w = (_lib.Window * max(n, 1))()
    for i, (r, a, b) in enumerate(windows):
        w[i].record, w[i].start, w[i].end = r, a, b
    out = (_lib.NestedHit * max(n, 1))()
    _lib.check(_lib.lib().ipcr_nested_windows(genome._h, w, n, inner._h, inner_scratch._h, out))
    return _convert(out, n, inner, None)


def NestedProducts(outer_scratch: engine.SimulationScratch, products: Sequence[engine.Product], genome: engine.Genome,
                   inner: engine.CompiledPanel, inner_scratch: engine.SimulationScratch,
                   require_inner: bool = False) -> List[NestedProduct]:
    """visitors.Nested.Visit over every product of the last scan on `outer_scratch` (`products` = what that scan
    returned, same order); require_inner drops outer products without an inner one (nested.go:23-26)"""
    n = len(products)
    out = (_lib.NestedHit * max(n, 1))()
    _lib.check(_lib.lib().ipcr_nested_products(outer_scratch._h, genome._h, inner._h, inner_scratch._h, out, n))
    res = _convert(out, n, inner, products)
    return [r for r in res if r.InnerFound] if require_inner else res
