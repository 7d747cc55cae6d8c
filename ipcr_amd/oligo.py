"""Mirror of core/oligo: probe validation and BestHit (the ipcr-probe amplicon rescan)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

from . import _lib


@dataclass
class Hit:
    """oligo.Hit -- core/oligo/oligo.go:9-15"""
    Found: bool = False
    Strand: str = ""
    Pos: int = 0
    MM: int = 0
    Site: str = ""


def Normalize(raw: str) -> str:
    """core/oligo/validate.go (same rule as core/primer/validate.go:12-22)"""
    return "".join(ch.upper() for ch in raw if not ch.isspace() and ch not in "'\"")


def Validate(raw: str) -> str:
    """core/oligo/validate.go:38-61: normalised IUPAC DNA or ValueError."""
    s = Normalize(raw)
    if not s:
        raise ValueError("empty oligo")
    for i, ch in enumerate(s):
        if ch not in "ACGTRYSWKMBDHVN":
            raise ValueError(f"invalid oligo base {ch!r} at position {i + 1}; "
                             "allowed: A C G T R Y S W K M B D H V N")
    return s


def BestHit(amplicon, probe: str, maxMM: int) -> Hit:
    """oligo.BestHit -- core/oligo/oligo.go:19-77, run on the device."""
    amp = amplicon if isinstance(amplicon, (bytes, bytearray)) else amplicon.encode()
    if not probe.strip():
        return Hit()
    prb = Validate(probe)  # the reference panics here (oligo.go:25-28)
    out = _lib.ProbeHit()
    _lib.check(_lib.lib().ipcr_probe_best_hit(bytes(amp), len(amp), prb.encode(), maxMM, C.byref(out)))
    if not out.found:
        return Hit()
    up = bytes(amp).upper()
    end = out.pos + len(prb)
    site = up[out.pos:end].decode() if end <= len(up) else ""
    return Hit(True, chr(out.strand), out.pos, out.mm, site)
