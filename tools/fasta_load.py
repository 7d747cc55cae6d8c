#!/usr/bin/env python3
"""FASTA loader rate from this interpreter, without torch being imported (development tool).

    python tools/fasta_load.py make /tmp/x.fa [bases] [records]    # synthetic upper-case FASTA, 80-column lines
    python tools/fasta_load.py load /tmp/x.fa [repeats]

IPCR_HIP_RUNTIME=system: bind libipcr_hip.so to /opt/rocm's HIP runtime instead of the one a PyTorch wheel bundles."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(path, bases=1_000_000_000, records=8):
    import numpy as np
    rng = np.random.default_rng(7)
    per = bases // records
    with open(path, "wb") as fh:
        for r in range(records):
            fh.write(b">chr%d synthetic\n" % (r + 1))
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, per, dtype=np.uint8)]
            full = (per // 80) * 80
            body = np.empty((full // 80, 81), dtype=np.uint8)
            body[:, :80] = seq[:full].reshape(-1, 80)
            body[:, 80] = 10
            fh.write(body.tobytes())
            if full < per:
                fh.write(seq[full:].tobytes() + b"\n")


def load(path, reps=3):
    from ipcr_amd import engine
    size = os.path.getsize(path)
    for r in range(reps):
        t0 = time.perf_counter()
        g = engine.Genome(size + (1 << 20), max_records=4096)
        g.add_fasta(path)
        s = time.perf_counter() - t0
        print("python load %d: %d records, %d bases, %.2f ms, %.2f GB/s of file, %.2f Gbases/s"
              % (r, g.num_records, g.total_bases, s * 1e3, size / s / 1e9, g.total_bases / s / 1e9))
        g.close()


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2], *[int(x) for x in sys.argv[3:5]])
    else:
        load(sys.argv[2], *[int(x) for x in sys.argv[3:4]])
