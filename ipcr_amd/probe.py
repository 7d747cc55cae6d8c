"""Mirror of core/probe: AnnotateAmplicon (core/probe/annotate.go:6-17)."""
from __future__ import annotations

from dataclasses import dataclass

from . import oligo


@dataclass
class Annotation:
    Found: bool = False
    Strand: str = ""
    Pos: int = 0
    MM: int = 0
    Site: str = ""


def AnnotateAmplicon(amplicon, probe: str, maxMM: int) -> Annotation:
    h = oligo.BestHit(amplicon, probe, maxMM)
    return Annotation(h.Found, h.Strand, h.Pos, h.MM, h.Site)
