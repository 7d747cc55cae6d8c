"""ipcr_amd -- MI355X-native seeded-scan + per-hit-verify primer matcher of ipcr.

Only the hot path (core/engine, core/primer, core/oligo of KPU-AGC/ipcr) lives here:
  csrc/      hand-written gfx950 kernels, host runtime, C ABI (include/ipcr_hip.h)
  engine.py  engine.New / CompilePanel / SimulateBatch / ForEachCompiledProduct mirror
  primer.py  primer.Pair, RevComp, self-pair rules
  oligo.py, probe.py  BestHit / AnnotateAmplicon (ipcr-probe rescan)
  dist.py    one process per GPU, genomes sharded over ranks, all-gatherv of hit records (RCCL)
"""
from . import engine, oligo, primer, probe  # noqa: F401

__all__ = ["engine", "primer", "oligo", "probe"]
