"""Mirror of the reference's core/primer package for the scan path (types + input helpers).

Host-side only: IUPAC tables and reverse complement come from the C ABI so that Python, the
cgo shim and the kernels share one definition.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

from . import _lib


@dataclass
class Pair:
    """primer.Pair -- core/primer/pair.go:4-10"""
    ID: str
    Forward: str
    Reverse: str
    MinProduct: int = 0
    MaxProduct: int = 0


@dataclass
class Oligo:
    """primer.Oligo -- core/primer/self.go:5-8"""
    ID: str
    Seq: str


@dataclass
class Match:
    """primer.Match -- core/primer/match.go:8-13"""
    Pos: int
    Mismatches: int
    Length: int
    MismatchIdx: Tuple[int, ...]


def BaseMatch(g: str, p: str) -> bool:
    """core/primer/iupac.go:62-67"""
    return bool(_lib.lib().ipcr_base_match(ord(g), ord(p)))


def IUPACMask(c: str) -> int:
    """core/primer/iupac.go:6-58"""
    return _lib.lib().ipcr_iupac_mask(ord(c))


def RevComp(seq) -> bytes:
    """core/primer/rc.go:26-56; raises (the reference panics) on anything but upper-case IUPAC."""
    b = seq if isinstance(seq, (bytes, bytearray)) else seq.encode()
    if not b:
        return b""
    import ctypes as C
    out = C.create_string_buffer(len(b))
    _lib.check(_lib.lib().ipcr_revcomp(bytes(b), len(b), out))
    return out.raw[:len(b)]


def Normalize(raw: str) -> str:
    """core/primer/validate.go:12-22"""
    return "".join(ch.upper() for ch in raw if not ch.isspace() and ch not in "'\"")


def Validate(raw: str) -> str:
    """core/primer/validate.go:27-37; ValueError where the reference returns an error."""
    s = Normalize(raw)
    if not s:
        raise ValueError("empty primer")
    for i, ch in enumerate(s):
        if ch not in "ACGTRYSWKMBDHVN":
            raise ValueError(f"invalid primer base {ch!r} at position {i + 1}; "
                             "allowed: A C G T R Y S W K M B D H V N")
    return s


def SelfPairs(oligos: Sequence[Oligo]) -> List[Pair]:
    """core/primer/self.go:13-25"""
    return [Pair(o.ID + "+self", o.Seq, o.Seq, 0, 0) for o in oligos]


def AddSelfPairs(pairs: Sequence[Pair]) -> List[Pair]:
    """internal/common/primers.go:11-37 (what `ipcr --self`, the default, scans)."""
    out = list(pairs)
    for p in pairs:
        if p.Forward:
            u = p.Forward.upper()
            out.append(Pair(p.ID + "+A:self", u, u, 0, 0))
        if p.Reverse:
            u = p.Reverse.upper()
            out.append(Pair(p.ID + "+B:self", u, u, 0, 0))
    return out


def AddSelfPairsUnique(pairs: Sequence[Pair]) -> List[Pair]:
    """internal/common/primers.go:41-74 (what ipcr-multiplex scans)."""
    out = list(pairs)
    seen_a, seen_b = set(), set()
    for p in pairs:
        f = p.Forward.strip().upper()
        if f and f not in seen_a:
            seen_a.add(f)
            out.append(Pair(p.ID + "+A:self", f, f, 0, 0))
        r = p.Reverse.strip().upper()
        if r and r not in seen_b:
            seen_b.add(r)
            out.append(Pair(p.ID + "+B:self", r, r, 0, 0))
    return out
