"""Synthetic workloads of BASELINE.json, restated from the reference's
benchmark fixture generators (core/engine/performance_benchmark_test.go:20-106)."""
from __future__ import annotations

from typing import List

from . import primer

_M32 = 0xFFFFFFFF


def bench_primer(idx: int, n: int = 20) -> str:
    """benchPrimer -- performance_benchmark_test.go:78-93"""
    x = (0x9e3779b9 ^ (idx * 0x45d9f3b)) & _M32
    buf = []
    for i in range(n):
        x = (x * 1103515245 + 12345 + i * 97) & _M32
        buf.append("ACGT"[(x >> 29) & 3])
    buf[0] = "ACGT"[idx & 3]
    buf[1] = "ACGT"[(idx + 1) & 3]
    buf[2] = "ACGT"[(idx + 2) & 3]
    buf[n - 1] = "ACGT"[(idx + 3) & 3]
    return "".join(buf)


def bench_pair(i: int) -> primer.Pair:
    """pair i of makeEngineBenchFixture -- performance_benchmark_test.go:27-46"""
    return primer.Pair("bench_%03d" % i, bench_primer(2 * i), bench_primer(2 * i + 1), 128, 212)


def c2_pairs() -> List[primer.Pair]:
    """C2: pair 0 as `ipcr` scans it with the default --self (internal/app/app.go:100-102)."""
    return primer.AddSelfPairs([bench_pair(0)])


def c3_pairs() -> List[primer.Pair]:
    """C3: 27F/1492R (README.md:46) with --self."""
    return primer.AddSelfPairs([primer.Pair("16S", "AGAGTTTGATCMTGGCTCAG", "TACGGYTACCTTGTTAYGACTT", 0, 0)])


def c4_pairs(n: int = 1024) -> List[primer.Pair]:
    """C4: n-pair TSV panel through ipcr-multiplex's unique self-pair rule
    (internal/multiplexapp/app.go:206-208, internal/common/primers.go:41-74)."""
    return primer.AddSelfPairsUnique([bench_pair(i) for i in range(n)])


def different_base(b: str) -> str:
    """performance_benchmark_test.go:95-106"""
    return {"A": "C", "C": "G", "G": "T"}.get(b, "A")
