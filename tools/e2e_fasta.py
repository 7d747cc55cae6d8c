#!/usr/bin/env python3
"""End-to-end timing SURVEY section 8(d)(iii) asks for: FASTA file -> resident tiles -> scan -> sorted TSV rows.

    python tools/e2e_fasta.py [--records 24] [--record-len 125000000] [--gz]

Writes a synthetic FASTA (the C2 genome: LCG records with planted amplicons, line width 80, upper case) under
$TMPDIR, then times every stage separately: file read + H2D + device normalisation + pack (ipcr_genome_add_fasta),
the scan (one launch over all records), and formatting + sorting the TSV rows (internal/common/sort.go order).
Development tool: numbers go to DESIGN.md section 5, never into bench.py's `value`."""
import argparse
import gzip
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=24)
    ap.add_argument("--record-len", type=int, default=125_000_000)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bench
    from ipcr_amd import cli, engine, primer, workloads

    def revcomp(s):
        return primer.RevComp(s)

    # the bench genome, copied back to the host record by record and written as 80-column FASTA
    tmpdir = os.environ.get("TMPDIR", "/tmp")
    path = os.path.join(tmpdir, "ipcr_e2e_%d.fa" % os.getpid()) + (".gz" if args.gz else "")
    t0 = time.perf_counter()
    genome, plants, _ = bench.build_genome(torch, engine, workloads, revcomp, 0, args.records, args.record_len, False)
    opener = (lambda p: gzip.open(p, "wb", compresslevel=1)) if args.gz else (lambda p: open(p, "wb"))
    with opener(path) as fh:
        for r in range(args.records):
            n = genome.record_len(r)
            seq = np.frombuffer(genome.read(r, 0, n), dtype=np.uint8)
            fh.write(b">chr%d synthetic LCG record\n" % (r + 1))
            full = (n // 80) * 80
            body = np.empty((full // 80, 81), dtype=np.uint8)
            body[:, :80] = seq[:full].reshape(-1, 80)
            body[:, 80] = 10
            fh.write(body.tobytes())
            if full < n:
                fh.write(seq[full:].tobytes() + b"\n")
    genome.close()
    fsize = os.path.getsize(path)
    t_write = time.perf_counter() - t0

    cfg = engine.Config(MaxMM=2, TerminalWindow=5, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12)
    eng = engine.New(cfg)
    cp = eng.CompilePanel(workloads.c2_pairs())
    sc = eng.NewSimulationScratch(cp)
    warm = engine.Genome(1 << 20, max_records=2)       # kernel specialisation (hiprtc) outside the timed stages
    warm.add_record("w", b"ACGT" * 1000)
    eng.ScanGenomeCount(warm, cp, sc)
    cp.wait_ready()
    warm.close()

    res = {"file_bytes": fsize, "gz": args.gz, "records": args.records, "bases": args.records * args.record_len,
           "host_cores": len(os.sched_getaffinity(0)), "write_fixture_s": round(t_write, 2)}
    for attempt in ("cold", "warm"):                   # second pass: file in the page cache
        t0 = time.perf_counter()
        g = engine.Genome(args.records * args.record_len + (1 << 20), max_records=args.records + 4)
        g.add_fasta(path)
        t1 = time.perf_counter()
        prods = eng.ScanGenome(g, cp, sc)
        t2 = time.perf_counter()
        rows = [(path, p) for p in prods]
        rows.sort(key=lambda t: cli.product_sort_key(t[0], t[1]))
        out = io.StringIO()
        out.write(cli.TSV_HEADER + "\n")
        for f, p in rows:
            out.write(cli.format_row(f, p) + "\n")
        t3 = time.perf_counter()
        res[attempt] = {"load_s": round(t1 - t0, 4), "load_GBps_file": round(fsize / (t1 - t0) / 1e9, 2),
                        "pack_ms": round(g.pack_ms, 2), "scan_ms": round((t2 - t1) * 1e3, 3),
                        "rows_ms": round((t3 - t2) * 1e3, 3), "products": len(prods),
                        "total_s": round(t3 - t0, 4), "gbases_per_s_end_to_end": round(g.total_bases / (t3 - t0) / 1e9, 2)}
        assert g.num_records == args.records and g.total_bases == args.records * args.record_len
        found = {(p.Record, p.Start) for p in prods if p.ExperimentID == "bench_000" and p.Type == "forward"}
        assert all((r, s) in found for (r, s, _) in plants), "planted amplicon missing after the FASTA round trip"
        g.close()
    if not args.keep:
        os.unlink(path)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
