#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X-native ipcr primer scan.  Prints ONE JSON line on rank 0.

    python bench.py                          # 1 GPU, workload C2 (the headline), other workloads in config.other_workloads
    python bench.py --gpus 8                 # starts 8 ranks itself (torch.distributed.run as a child process)
    python bench.py --workload c4            # the 1024-row multiplex panel as the main line

Workloads (BASELINE.json configs[1..4]; synthetic 3.0 Gb genome per GPU = 24 records x 125 Mb
of the reference's benchDNA LCG, resident in HBM as 2-bit + invalid-bit tiles when the timed region starts):
  c2  one primer pair as `ipcr` scans it with the default --self (12 orientation slots, 4 distinct patterns), k=2,
      3'-window 5, hit-cap 10000, max-length 2000; 1000 planted 180-bp amplicons (exact / 1 / 2 mismatches)
  c3  27F/1492R (IUPAC M/Y), k=3, 3'-window 3, --circular; 480 planted amplicons + one origin-spanning per record
  c4  1024-row ipcr-multiplex panel (3072 pairs / 4096 distinct patterns), k=2, 3'-window 3: seed-index filter
  c5  c2's pair + an internal probe (ipcr-probe): scan + batched probe rescan of every product
One STEP = one pass of the hot path over the whole resident genome: sweep (filter, exact verification and hand-over)
-> hit records on the host -> reference-order match lists -> amplicon join -> products (c5: + probe rescan); the
product count of EVERY pass is checked, every planted amplicon of the last one.  Passes are pipelined over three
scratches (pipelined_passes); --no-pipeline runs them one by one.

N > 1 (one process per GPU): weak scaling -- every rank scans its own genome (seed + rank) with the same panel and
joins its own records; the hit records of every pass are exchanged by one all-gather over RCCL straight out of the
device hit buffer (two in flight); rank 0 joins the whole job from the gathered records of the last pass as a check.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_BASE = 0.375         # lo + hi + inv planes: the encoded tile a filter kernel reads once
RECORDS = 24
RECORD_LEN = 125_000_000
N_PLANTS = 1000
PRODUCT_LEN = 180
MIN_WARM_PASSES = {"c2": 300, "c2n": 300, "c3": 300, "c5": 300, "c4": 12, "c4n": 12}   # the first ~50 sweeps after idle run 15-25 % slower (clock ramp)
PROBE = "TGGACCTTAGCAGGTCATTCAG"


# ----------------------------------------------------------------------------------------------- launch
def launch_plan(gpus: int, env, argv, python: str = sys.executable, script: str = os.path.abspath(__file__)):
    """How `bench.py --gpus N` gets its N ranks (one process per GPU, internal/pipeline/pipeline.go:60-125 is the
    reference's unit of parallelism: independent records over a pool of workers).
      * RANK in the environment: this process IS a rank (torchrun or the driver started it) -> None = run here;
        WORLD_SIZE must then equal --gpus when --gpus > 1 was given.
      * no RANK and N > 1: this process only launches -- it returns the command of
        `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`, which main() starts as a CHILD
        process (never exec) before anything has touched the GPU, and whose exit code it returns.
      * N <= 1: None."""
    if "RANK" in env:
        world = int(env.get("WORLD_SIZE", "1"))
        if gpus > 1 and world != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: start one rank per GPU "
                             f"(python bench.py --gpus {gpus} does it itself)")
        return None
    if gpus <= 1:
        return None
    port = env.get("MASTER_PORT") or str(29500 + (os.getpid() % 400))
    return [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", port, script] + list(argv)


def pipelined_passes(k, scratches, begin, end, chain=None, start_exchange=None, finish_exchange=None, pipeline=True,
                     recv_slots=2):
    """k passes of the hot path over `scratches` in rotation, pipelined the way the reference's worker pool +
    collector are (internal/pipeline/pipeline.go:60-161): while the host waits for and joins pass i, the sweep of pass
    i+1 is already queued (chain(cur, prev): behind pass i's on the device); with several GPUs the all-gather of pass
    i's hit records runs under the following passes.  Invariants (tests/test_host_logic.py checks them with fakes):
      * a scratch is begun again only after its previous pass has been ended AND the exchange that reads its device
        hit buffer has finished;
      * at most `recv_slots` exchanges are in flight (the exchanger has that many receive buffers);
      * when the function returns every pass has been ended and every exchange finished.
    Returns what end() returned for the last pass."""
    n, works, ns = None, {}, len(scratches)   # works: pass -> exchange in flight (it reads that pass's scratch)
    if k <= 0:
        return n

    def do_begin(j):
        if finish_exchange is not None:
            for jj in [q for q in works if q <= j - ns]:
                finish_exchange(works.pop(jj))
        if j > 0 and pipeline and chain is not None:
            chain(scratches[j % ns], scratches[(j - 1) % ns])
        begin(scratches[j % ns])

    do_begin(0)
    for i in range(k):
        cur = scratches[i % ns]
        if i + 1 < k and pipeline:
            do_begin(i + 1)
        n = end(i, cur)
        if start_exchange is not None:
            for jj in [q for q in works if q <= i - recv_slots]:
                finish_exchange(works.pop(jj))
            works[i] = start_exchange(cur)
        if i + 1 < k and not pipeline:
            do_begin(i + 1)
    for jj in sorted(works):
        finish_exchange(works.pop(jj))
    return n


# ----------------------------------------------------------------------------------------------- genomes
class Ctx:
    """everything a workload run needs: modules, rank layout, the exchanger's device, arguments"""
    pass


def _put(torch, buf, pos, text):
    b = text if isinstance(text, (bytes, bytearray)) else text.encode()
    buf[pos:pos + len(b)] = torch.tensor(list(b), dtype=torch.uint8)


def plant_stride(records: int, record_len: int) -> int:
    per_rec = (N_PLANTS + records - 1) // records
    return max((record_len - 4096) // (per_rec + 1), 400)


def n_runs(record_len: int, genome_idx: int, record: int, keep_clear):
    """SURVEY.md 8(d), variant +N: 0.1 % of a record's positions overwritten with 'N' in runs of 1..1000 (own xorshift32,
    fixed seed per genome and record) -- the reference's forceFallback condition for every record (core/engine/compiled.go:185-190)
    and its halo path (halo.go:76-108).  keep_clear: sorted (lo, hi) windows no run may touch (the planted amplicons: the
    same plants are verified as in the N-free genome).  -> [(start, length)], in generation order."""
    import bisect
    x = (0x2545F491 ^ ((genome_idx * 0x9E3779B9 + record * 0x85EBCA6B) & 0xFFFFFFFF)) or 1
    los = [w[0] for w in keep_clear]
    runs, covered, target = [], 0, record_len // 1000
    while covered < target:
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        ln = 1 + x % 1000
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        if record_len <= ln + 1:
            break
        pos = x % (record_len - ln)
        i = bisect.bisect_right(los, pos + ln)
        if i > 0 and keep_clear[i - 1][1] > pos:
            continue
        runs.append((pos, ln))
        covered += ln
    return runs


def build_genome(torch, engine, workloads, revcomp_fn, genome_idx: int, records: int, record_len: int,
                 keep_host_record0: bool, probe: str = "", with_n: bool = False):
    """C2 / C4 / C5 genome of SURVEY.md 8(d): LCG stream seed 0x5eed1234+g cut into records, amplicons of pair 0
    planted as makeEngineBenchFixture does (performance_benchmark_test.go:47-62); with `probe` (C5) every amplicon
    also carries the probe at offset 60: as is, reverse-complemented, with 1 or 2 substitutions, or not at all."""
    from ipcr_amd import workloads as W
    g = engine.Genome(records * record_len, records)
    pair = W.bench_pair(0)
    fwd = pair.Forward
    rc_rev = revcomp_fn(pair.Reverse)
    rc_probe = revcomp_fn(probe) if probe else b""
    buf = torch.empty(record_len, dtype=torch.uint8, device="cuda")
    plants = []  # (record, start, n_mismatches)
    per_rec = (N_PLANTS + records - 1) // records
    stride = plant_stride(records, record_len)
    host0 = None
    for r in range(records):
        engine.lcg_fill_device(buf.data_ptr(), record_len, 0x5eed1234 + genome_idx, r * record_len)
        for t in range(per_rec):
            gidx = t * records + r
            if gidx >= N_PLANTS:
                break
            start = 2048 + t * stride
            if start + PRODUCT_LEN + 64 > record_len:
                break
            nm = gidx % 3
            site = list(fwd)
            if nm >= 1:
                site[10] = W.different_base(site[10])
            if nm >= 2:
                site[3] = W.different_base(site[3])
            _put(torch, buf, start, "".join(site))
            _put(torch, buf, start + PRODUCT_LEN - 20, rc_rev)
            if probe:
                kind = gidx % 5          # 0 '+', 1 '-', 2 one substitution, 3 two, 4 no probe site
                ps = list(probe)
                if kind in (2, 3):
                    ps[5] = W.different_base(ps[5])
                if kind == 3:
                    ps[15] = W.different_base(ps[15])
                if kind != 4:
                    _put(torch, buf, start + 60, rc_probe if kind == 1 else "".join(ps))
            plants.append((r, start, nm))
        if with_n:
            clear = sorted((s0 - 64, s0 + PRODUCT_LEN + 64) for (rr, s0, _) in plants if rr == r)
            for (pos, ln) in n_runs(record_len, genome_idx, r, clear):
                buf[pos:pos + ln] = 78   # 'N'
        torch.cuda.synchronize()
        if keep_host_record0 and r == 0:
            host0 = buf.cpu().numpy().copy()
        g.add_record_device("chr%d" % (r + 1), buf.data_ptr(), record_len)
    del buf
    return g, plants, host0


def build_genome_c3(torch, engine, workloads, revcomp_fn, genome_idx: int, records: int, record_len: int,
                    keep_host_record0: bool = False):
    """C3 genome: LCG records (seed 0x5eed3333 + g), 27F / rc(1492R) planted 400 bp apart with concrete bases for the
    IUPAC codes (M -> A/C, Y -> C/T alternating) and 0..3 substitutions outside the 3' window; one amplicon per record
    spans the origin (forward site near the end, reverse site near the start: found in --circular mode only)."""
    from ipcr_amd import workloads as W
    pair = W.c3_pairs()[0]
    fwd, rev = pair.Forward, pair.Reverse
    rc_rev = revcomp_fn(rev).decode()
    opts = {"A": "A", "C": "C", "G": "G", "T": "T", "M": "AC", "Y": "CT", "R": "AG", "K": "GT"}

    def concrete(s, salt):
        return "".join(opts[ch][(salt + i) % len(opts[ch])] for i, ch in enumerate(s))

    g = engine.Genome(records * record_len, records)
    buf = torch.empty(record_len, dtype=torch.uint8, device="cuda")
    host0 = None
    plants = []  # (record, start, mismatch idx tuple)
    per_rec = 20
    stride = max((record_len - 2_000_000) // (per_rec + 1), 1000)
    for r in range(records):
        engine.lcg_fill_device(buf.data_ptr(), record_len, 0x5eed3333 + genome_idx, r * record_len)
        for t in range(per_rec):
            start = 1_000_000 + t * stride
            if start + 464 > record_len - 1000:
                break
            site = list(concrete(fwd, r + t))
            want = [j for j in ((2, 7, 12)[: (r + t) % 4]) if fwd[j] in "ACGT"]     # 0..3 substitutions, 3' window clean
            for j in want:
                site[j] = W.different_base(site[j])
            _put(torch, buf, start, "".join(site))
            _put(torch, buf, start + 400 - len(rc_rev), concrete(rc_rev, r))
            plants.append((r, start, tuple(want)))
        if record_len > 4000:
            _put(torch, buf, record_len - 150, concrete(fwd, r))
            _put(torch, buf, 100, concrete(rc_rev, r))
        torch.cuda.synchronize()
        if keep_host_record0 and r == 0:
            host0 = buf.cpu().numpy().copy()
        g.add_record_device("chr%d" % (r + 1), buf.data_ptr(), record_len)
    del buf
    return g, plants, host0


# ----------------------------------------------------------------------------------------------- one workload
def workload_spec(name, engine, workloads):
    E = engine
    if name == "c2":
        return dict(cfg=E.Config(MaxMM=2, TerminalWindow=5, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12),
                    pairs=workloads.c2_pairs(), genome="c2", kernel="ipcr_filter",
                    text="C2: 1 primer pair (+self pairs: 12 orientation slots, 4 distinct patterns), k=2, 3'-window=5, "
                         "hit-cap 10000, max-length 2000")
    if name == "c2n":
        return dict(cfg=E.Config(MaxMM=2, TerminalWindow=5, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12),
                    pairs=workloads.c2_pairs(), genome="c2n", kernel="ipcr_filter",
                    text="C2 on the +N genome (SURVEY 8d: 0.1 % of positions in runs of 1-1000 N): every record holds reset bytes, "
                         "so with k > 0 and a hit cap the reference takes FindMatches for every orientation "
                         "(core/engine/compiled.go:185-190,238-258) and caps the rc orientations before their 5' window filter -- "
                         "the device scans those two patterns unprotected and the host filters (the kernel every record and "
                         "every chunk with a non-ACGT byte runs, and every chunk the device packs)")
    if name == "c3":
        return dict(cfg=E.Config(MaxMM=3, TerminalWindow=3, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12, Circular=True),
                    pairs=workloads.c3_pairs(), genome="c3", kernel="ipcr_filter",
                    text="C3: 27F/1492R with IUPAC codes M/Y (+self pairs: 4 distinct patterns of 20 and 22 nt), k=3, "
                         "3'-window=3, --circular, hit-cap 10000, max-length 2000")
    if name == "c4":
        return dict(cfg=E.Config(MaxMM=2, TerminalWindow=3, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12),
                    pairs=workloads.c4_pairs(1024), genome="c2", kernel="ipcr_index_filter",
                    text="C4: 1024-row ipcr-multiplex panel (unique self pairs added: 3072 pairs, 4096 distinct patterns), "
                         "k=2, 3'-window=3, hit-cap 10000, max-length 2000")
    if name == "c4n":
        return dict(cfg=E.Config(MaxMM=2, TerminalWindow=3, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12),
                    pairs=workloads.c4_pairs(1024), genome="c2n", kernel="ipcr_index_filter",
                    text="C4 on the +N genome: the 1024-row panel where every record holds reset bytes -- the rc orientations are scanned "
                         "without their 5' window (the reference caps them before the window filter), filed under split keys: "
                         "window exact + one of k+1 blocks, or a mismatch in the window + one of k longer blocks")
    if name == "c5":
        return dict(cfg=E.Config(MaxMM=2, TerminalWindow=5, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12),
                    pairs=workloads.c2_pairs(), genome="c5", kernel="ipcr_filter", probe=PROBE,
                    text="C5: ipcr-probe = C2's pair + internal probe %s (--probe-max-mm 2): scan, then the batched probe "
                         "rescan of every product" % PROBE)
    raise SystemExit("unknown workload " + name)


def get_genome(ctx, key):
    """genomes are built once per process and shared by the workloads that scan them"""
    if key in ctx.genomes:
        return ctx.genomes[key]
    torch, engine, workloads = ctx.torch, ctx.engine, ctx.workloads
    a = ctx.args
    if key == "c3":
        g, plants, host0 = build_genome_c3(torch, engine, workloads, ctx.revcomp, ctx.rank, a.records, a.record_len,
                                           keep_host_record0=ctx.want_cpu or ctx.want_cpu_others)
    else:
        want_host = key == "c2" and (ctx.want_cpu or ctx.want_cpu_others)
        g, plants, host0 = build_genome(torch, engine, workloads, ctx.revcomp, ctx.rank, a.records, a.record_len,
                                        want_host, probe=PROBE if key == "c5" else "", with_n=key == "c2n")
    nrec = g.num_records
    lens = [g.record_len(r) for r in range(nrec)]
    flags = [g.record_flags(r) for r in range(nrec)]
    all_lens, all_flags = ctx.dist.allgather_record_meta(lens, flags, device=ctx.cdev) if ctx.multi else (lens, flags)
    ctx.genomes[key] = dict(g=g, plants=plants, host0=host0, nrec=nrec, all_lens=all_lens, all_flags=all_flags)
    return ctx.genomes[key]


def check_products(name, prods, plants):
    """every planted amplicon of rank 0's genome must come back exactly (coordinates, mismatch counts and positions)"""
    if name in ("c2", "c2n", "c4", "c4n", "c5"):
        found = {(p.Record, p.Start): p for p in prods
                 if p.ExperimentID == "bench_000" and p.Type == "forward" and p.Length == PRODUCT_LEN}
        for (r, start, nm) in plants:
            p = found.get((r, start))
            assert p is not None, f"{name}: planted amplicon missing: record {r} start {start}"
            want_idx = () if nm == 0 else ((10,) if nm == 1 else (3, 10))
            assert p.FwdMM == nm and p.FwdMismatchIdx == want_idx and p.RevMM == 0, (name, p, nm)
    else:
        found = {(p.Record, p.Start): p for p in prods if p.ExperimentID == "16S" and p.Type == "forward" and p.Length == 400}
        for (r, start, idx) in plants:
            p = found.get((r, start))
            assert p is not None, f"c3: planted amplicon missing: record {r} start {start}"
            assert (p.FwdMM, p.FwdMismatchIdx, p.RevMM) == (len(idx), idx, 0), (p, idx)
        wraps = [p for p in prods if p.ExperimentID == "16S" and p.Type == "forward" and p.Start > p.End]
        recs = {r for (r, _, _) in plants}
        assert {p.Record for p in wraps} >= recs, "c3: an origin-spanning amplicon is missing (circular join)"


def run_workload(ctx, name, steps, warmup):
    """Set up workload `name`, run `warmup` untimed and `steps` timed passes (barrier + synchronize on both sides,
    MAX over ranks), verify the last pass.  Returns a dict of measurements."""
    torch, engine, dist, tdist = ctx.torch, ctx.engine, ctx.dist, ctx.tdist
    _lib = ctx._lib
    spec = workload_spec(name, engine, ctx.workloads)
    G = get_genome(ctx, spec["genome"])
    genome, nrec = G["g"], G["nrec"]
    t_setup = time.perf_counter()
    eng = engine.New(spec["cfg"])
    # time to first product of a panel: CompilePanel (core/engine/compiled.go:96-136; the reference benchmarks it on its
    # own, performance_benchmark_test.go:108-153) + the first scan, which builds the panel's kernel with hiprtc -- COLD: the
    # code-object cache directory is empty for this source; WARM: another panel object of the same source, the code object
    # read back from the disk cache (what a second `ipcr` run with the same primers pays)
    compile_times = panel_compile_times(ctx, eng, spec, genome) if name not in ctx.compile_done else None
    ctx.compile_done.add(name)
    cp = eng.CompilePanel(spec["pairs"])
    # three scratches in rotation: one being swept, one being joined on the host, one whose hit buffer the
    # all-gather of the pass before may still be reading (several GPUs); two would do on one GPU
    scs = [eng.NewSimulationScratch(cp) for _ in range(3)]
    multi = ctx.multi
    xchg, host_sc = None, None
    if multi:
        host_sc = engine.SimulationScratch(cp, host_only=True)
        xchg = dist.HitExchanger(device=ctx.cdev)
        eng.ScanGenomeHits(genome, cp, scs[0])      # one synchronous exchange: sizes the buffers on every rank
        xchg.allgather(dist.hits_from_scratch(scs[0]), nrec, size_hint=int(scs[0].device_hits()[1] * 1.25) + 64)
        xchg.agree_on_device_path(scs[0])           # zero-copy view of the device hit buffer on every rank, or the host copy on all
        # the library's own exchange against the torch.distributed form of the same hits, once, on every rank; all ranks fall
        # back together if they differ anywhere (dist.py: verify_native) -- the first job with a real multi-rank communicator
        # checks the path that one-GPU boxes can only run with one rank
        xchg.verify_native(scs[0], nrec)
    expect = None
    for s_ in scs:  # untimed set-up: kernel specialisation (hiprtc) and buffer sizing happen here
        n_ = eng.ScanGenomeCount(genome, cp, s_)
        assert expect in (None, n_)
        expect = n_
        cp.wait_ready()   # (a small panel's kernels are built in the background: the measured passes run on them)
    setup_s = time.perf_counter() - t_setup
    probe = spec.get("probe")
    probe_out = (_lib.ProbeHit * max(expect, 1))() if probe else None
    probe_ms = []

    def run_steps(k):
        fms = []

        def end(i, cur):
            n = eng.ScanGenomeEndCount(genome, cp, cur)          # this rank's partition of the join: its records
            if n != expect:                                      # every pass is checked, not only the last one
                raise SystemExit(f"{name} pass {i}: {n} products, the set-up scan found {expect}")
            if probe:                                            # ipcr-probe: rescan every product's amplicon (internal/visitors/probe.go:18-33)
                t0 = time.perf_counter()
                _lib.check(_lib.lib().ipcr_probe_products(cur._h, genome._h, probe.encode(), 2, probe_out, n))
                probe_ms.append((time.perf_counter() - t0) * 1e3)
            fms.append(cur.stats().filter_ms)
            return n

        n = pipelined_passes(
            k, scs,
            begin=lambda cur: eng.ScanGenomeBegin(genome, cp, cur),
            end=end,
            chain=None if (ctx.args.no_pipeline or os.environ.get("IPCR_BENCH_NO_CHAIN")) else (lambda cur, prev: cur.chain_after(prev)),
            start_exchange=(lambda cur: xchg.start_scratch(cur, nrec)) if multi else None,   # all-gatherv of hit records, async
            finish_exchange=xchg.finish if multi else None,
            pipeline=not ctx.args.no_pipeline)
        return fms, n

    # whatever --warmup the caller picks, the device has done at least MIN_WARM_PASSES before the timed region
    # (reported as warmup_actual): the first ~50 sweeps after idle run 15-25 % slower (clock ramp)
    hidden = max(0, MIN_WARM_PASSES[name] - max(warmup, 0))
    run_steps(hidden)
    run_steps(max(warmup, 0))

    def timed_window():
        """EXACTLY `steps` passes between barrier + synchronize on both sides -> (seconds, sweep times, products)"""
        if multi:
            tdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fms, n = run_steps(steps)
        torch.cuda.synchronize()
        if multi:
            tdist.barrier()
        return time.perf_counter() - t0, fms, n

    # A window of few short passes (the driver's 20 x 0.19 ms = 4 ms) is thin evidence: such a window is repeated -- every
    # repetition again exactly `steps` passes, bracketed the same way -- and the MEDIAN window is reported (all of them
    # in window_ms_min / max).  Long windows (>= 50 ms) stand alone.
    el0, filter_ms, nprod = timed_window()
    windows = [el0]
    if el0 < 0.05 and not os.environ.get("IPCR_BENCH_ONE_WINDOW"):
        while len(windows) < 25:
            el, fms, n = timed_window()
            windows.append(el)
            filter_ms += fms
            nprod = n
    if multi:  # every rank ran the same number of windows: the slowest rank's time, window by window
        tw = torch.tensor(windows + [0.0] * (25 - len(windows)), dtype=torch.float64, device=ctx.cdev)
        tdist.all_reduce(tw, op=tdist.ReduceOp.MAX)
        windows = [float(x) for x in tw.cpu()[:len(windows)]]
        ptot = torch.tensor([nprod], dtype=torch.int64, device=ctx.cdev)
        tdist.all_reduce(ptot, op=tdist.ReduceOp.SUM)
        nprod = int(ptot.item())
    elapsed = sorted(windows)[len(windows) // 2]
    last = scs[(steps - 1) % len(scs)]   # scratch holding the last pass
    # ---- correctness outside the timed region ----
    if not multi:
        prods = last.products(genome.ids)
    else:  # rank 0 joins the WHOLE job from the gathered hits and checks its own genome's plants
        allhits, _, _ = xchg.gathered()                     # what the last step's all-gatherv left on every rank
        prods = eng.JoinHits(cp, host_sc, allhits, G["all_lens"], G["all_flags"]) if ctx.rank == 0 else []
        if ctx.rank == 0:
            assert len(prods) == nprod, f"{name}: whole-job join on rank 0 found {len(prods)} products, partitioned join {nprod}"
    if ctx.rank == 0:
        check_products(name, prods, G["plants"])            # rank 0's own genome occupies records [0, nrec)
        if probe and not multi:
            kinds = {0: ("+", 0), 1: ("-", 0), 2: ("+", 1), 3: ("+", 2)}
            byplant = {(p.Record, p.Start): i for i, p in enumerate(prods)
                       if p.ExperimentID == "bench_000" and p.Type == "forward" and p.Length == PRODUCT_LEN}
            stride = plant_stride(nrec, ctx.args.record_len)
            for (r, start, _nm) in G["plants"]:
                kind = (((start - 2048) // stride) * nrec + r) % 5
                h = probe_out[byplant[(r, start)]]
                if kind == 4:
                    continue    # no planted probe site: whatever the background holds
                assert h.found and (chr(h.strand), h.pos, h.mm) == (kinds[kind][0], 60, kinds[kind][1]), (r, start, kind, h.found, h.pos, h.mm)
    st = last.stats()
    total_bases = genome.total_bases * ctx.world
    fms_avg = sum(filter_ms) / len(filter_ms)
    res = dict(
        name=name, text=spec["text"], kernel=spec["kernel"], steps=steps, warmup=warmup, warmup_actual=hidden + max(warmup, 0),
        elapsed=elapsed, ms_per_step=elapsed * 1e3 / steps, value=total_bases * steps / elapsed / 1e9,
        filter_ms=fms_avg, nprod=int(nprod), hits=int(st.hits), candidates=int(st.candidates), kernel_kind=int(st.kernel_kind),
        n_patterns=int(st.n_patterns), setup_s=setup_s, bases_per_gpu=genome.total_bases, nrec=nrec, plants=len(G["plants"]),
        pack_ms=genome.pack_ms, breakdown={k: round(getattr(st, k), 4) for k in
                                           ("filter_ms", "verify_ms", "enqueue_ms", "wait_ms", "sort_ms", "join_ms", "total_ms")},
        device_path=bool(xchg.device_path) if multi else None, exchange_redone=xchg.redone if multi else 0,
        native_exchange=bool(xchg.native) if multi else None, native_exchange_verified=xchg.native_verified if multi else None,
        compile_times=compile_times,
        windows=len(windows), window_ms_min=min(windows) * 1e3, window_ms_max=max(windows) * 1e3,
        probe_ms=(sum(probe_ms[-steps:]) / max(1, len(probe_ms[-steps:]))) if probe_ms else None,
        prods=prods)
    for s_ in scs:
        s_.close()
    if host_sc is not None:
        host_sc.close()
    if xchg is not None:
        xchg.close()
    cp.close()
    return res


def panel_compile_times(ctx, eng, spec, genome):
    """-> {"compile_panel_s", "first_scan_cold_s", "first_scan_warm_s"}: see run_workload.  Single-rank jobs only (the
    several ranks of a node would race for the same cache files; their kernels are the same)."""
    if ctx.world > 1:
        return None
    import tempfile
    out = {}
    keep = os.environ.get("IPCR_JIT_CACHE_DIR")
    with tempfile.TemporaryDirectory(prefix="ipcr_bench_jit_") as d:
        os.environ["IPCR_JIT_CACHE_DIR"] = d
        os.environ["IPCR_JIT_NO_MEMCACHE"] = "1"      # nothing this process compiled earlier: what a fresh process sees
        try:
            for phase in ("cold", "warm"):
                t0 = time.perf_counter()
                cp = eng.CompilePanel(spec["pairs"])
                t1 = time.perf_counter()
                sc = eng.NewSimulationScratch(cp)
                eng.ScanGenomeCount(genome, cp, sc)
                t2 = time.perf_counter()
                kind = int(sc.stats().kernel_kind)
                cp.wait_ready()
                t3 = time.perf_counter()
                out["compile_panel_s"] = round(t1 - t0, 4)
                # products of the first scan of the whole genome are there after first_products_s; a small panel's first scans
                # run on the table-driven kernel (kind 2) while hiprtc builds its own in the background (kernels_built_s)
                out["first_products_%s_s" % phase] = round(t2 - t1, 4)
                out["first_scan_kernel_kind_%s" % phase] = kind
                out["kernels_built_%s_s" % phase] = round(t3 - t1, 4)
                sc.close()
                cp.close()
        finally:
            del os.environ["IPCR_JIT_NO_MEMCACHE"]
            if keep is None:
                del os.environ["IPCR_JIT_CACHE_DIR"]
            else:
                os.environ["IPCR_JIT_CACHE_DIR"] = keep
    return out


def roofline_of(res, traffic_file=None):
    alg_bytes = res["bases_per_gpu"] * BYTES_PER_BASE
    achieved = alg_bytes / (res["filter_ms"] * 1e-3) / 1e9
    out = {
        "bound": "hbm",
        "kernel": res["kernel"],
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None,
        "algorithmic_bytes_per_launch": int(alg_bytes),
        "frac_of_copy_peak": round(achieved / 6290.0, 4),  # 6.29 TB/s: what a float4 copy reaches (MI355X_MICROARCH.md)
        "avg_launch_ms": round(res["filter_ms"], 4),
        "gbases_per_s_kernel": round(res["bases_per_gpu"] / (res["filter_ms"] * 1e-3) / 1e9, 1),
    }
    # HBM traffic needs a rocprofv3 --pmc pass, which this process cannot run on itself: the number below is what the
    # PMC passes of the SAME command measured when profiles/ was last collected, and is labelled as such
    if traffic_file and os.path.exists(traffic_file):
        try:
            d = json.load(open(traffic_file))
            if d.get("kernel") == res["kernel"] and d.get("hbm_bytes_per_launch"):
                out["traffic"] = d["hbm_bytes_per_launch"]
                out["traffic_source"] = ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command when profiles/ "
                                         "was collected (FETCH_SIZE doubled, MI355X_MICROARCH.md HBM); not measured by this run"
                                         % os.path.relpath(traffic_file, ROOT))
        except Exception:
            pass
    if res["kernel"] == "ipcr_index_filter":
        lim = ("not HBM: the LDS first, VALU issue behind it (per 64-base step ~29 instructions of walk -- one random 1-byte "
               "bitmap lookup per key shape, 3.4 lanes of 32 on the busiest bank -- and ~11 of drain: ~0.12 key hits per lane and "
               "step are checked exactly, in rounds of 64 that the stack-ordered queue keeps full, and nearly all rejected); "
               "halving the lookups at +43 % instructions made the sweep 18 % slower; see DESIGN.md section 4.3")
        try:
            d = json.load(open(traffic_file))
            dv = d.get("derived")
            # the counters describe the run they were collected on: quoted only for the same genome size
            if dv and d.get("algorithmic_bytes_per_launch") == int(alg_bytes):
                lim += ("; %s when profiles/ was collected: LDS busy %.0f %% of the sweep (%.0f %% of it bank conflicts, %.1f LDS "
                        "instructions and %.1f VALU instructions per 64-base step)"
                        % (os.path.relpath(traffic_file, ROOT), 100 * dv["lds_busy_frac"], 100 * dv["lds_bank_conflict_frac_of_busy"],
                           dv["lds_instructions_per_base_step"], dv["valu_instructions_per_base_step"]))
        except Exception:
            pass
        out["limiter"] = lim
    return out


# ----------------------------------------------------------------------------------------------- other measurements
def scan_chunk_rates(args, local=0, record_bases=125_000_000, chunk=4_000_000, probe=False, with_n=False, panel_rows=0, workers=None):
    """Drop-in entry point (what the cgo shim binds): ipcr_scan_chunk on host ASCII under the reference's worker model
    (internal/pipeline/pipeline.go:60-125): W threads, one scratch each, one shared panel, rolling chunks of one
    record from a queue.  PCIe-inclusive; reported next to the raw pinned H2D rate; never `value`.  Measured by the
    native driver ipcr_amd/chunk_workers (csrc/chunk_workers.cpp) in a child process: the call takes ~0.1 ms, and a
    Python thread pool would add its own per-call interpreter work to it.  It runs BEFORE this process touches the GPU
    and with GPU_MAX_HW_QUEUES=4, the runtime's default and the best setting for a pool of packing workers (chunk_workers.cpp;
    this process itself runs with 16, one queue per stream, for the resident-genome lanes and the collective)."""
    import subprocess
    exe = os.path.join(ROOT, "ipcr_amd", "chunk_workers")
    if not os.path.exists(exe):
        raise SystemExit(exe + " is missing: build first (python -c 'import __graft_entry__ as g; g.build()')")
    n = min(record_bases, args.record_len)
    # probe: BASELINE C5 over the drop-in call -- every worker annotates its chunk's products (ipcr_probe_scratch_products),
    # a collector thread calls ipcr_probe_best_hit per amplicon beside them (chunk_workers.cpp: --probe)
    # with_n: 0.1 % of the record's positions are N (every chunk then holds a reset byte and takes the pattern set without the rc
    # orientations' window; a clean chunk packed by the host keeps it).  panel_rows: an n-row multiplex panel (k=2, window 3)
    # instead of C2's -- the seed-index kernel under the worker pool
    argv = [exe] + (["--probe"] if probe else []) + (["--with-n"] if with_n else []) + [str(n), str(chunk)]
    argv += [str(w) for w in (workers or ((16,) if probe else (1, 8, 16)))]
    env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local)),
               GPU_MAX_HW_QUEUES=os.environ.get("IPCR_CHUNK_HW_QUEUES", "4"))   # see chunk_workers.cpp
    if panel_rows:
        env["CHUNK_PANEL_ROWS"] = str(panel_rows)
    r = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=env)
    if r.returncode != 0:
        raise SystemExit("chunk_workers failed (%d): %s" % (r.returncode, r.stderr[-2000:]))
    return json.loads(r.stdout.strip().splitlines()[-1])


def scan_chunk_all_devices(world: int, args, record_bases=125_000_000, chunk=4_000_000):
    """Several GPUs: the ONE-process form of the drop-in path -- the native worker pool with its workers spread over every
    device of the job (chunk_workers --devices 0,..,N-1: worker i -> device i mod N through ipcr_scratch_create_on, no
    collective; internal/pipeline/pipeline.go:60-125).  Run by rank 0 in a child process before it touches the GPU;
    reported under other_workloads, never `value`.  Any failure is reported as text: it must not take the job down."""
    import subprocess
    exe = os.path.join(ROOT, "ipcr_amd", "chunk_workers")
    try:
        env = dict(os.environ, GPU_MAX_HW_QUEUES=os.environ.get("IPCR_CHUNK_HW_QUEUES", "4"))
        if os.environ.get("IPCR_BENCH_ONE_DEVICE"):      # rehearsal on one GPU: device slots instead of devices
            env["IPCR_DEVICE_SLOTS"] = str(world)
        r = subprocess.run([exe, "--devices", ",".join(str(d) for d in range(world)), str(min(record_bases, args.record_len)), str(chunk),
                            str(2 * world), str(4 * world)], capture_output=True, text=True, timeout=300, env=env)
        if r.returncode != 0:
            return {"error": "chunk_workers --devices failed (%d): %s" % (r.returncode, r.stderr[-500:])}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:500]}


def measure_traffic(workload: str, kernel: str):
    """HBM bytes per launch of the dominant kernel, measured IN THIS RUN: two rocprofv3 --pmc children (FETCH_SIZE and
    WRITE_SIZE in passes of their own, nothing else traced, the program directly after `--`) of this script on the same
    workload, started before this process touches the GPU.  FETCH_SIZE is reported in KiB and, on gfx950, at half the
    bytes of a 16 B/lane coalesced stream (MI355X_MICROARCH.md, HBM): read = raw KiB x 1024 x 2; WRITE_SIZE is exact.
    -> (bytes per launch, launches averaged) or None (no rocprofv3, a failed pass: the caller quotes profiles/ instead)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    vals = {}
    with tempfile.TemporaryDirectory(prefix="ipcr_bench_pmc_") as d:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(d, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--workload", workload, "--no-cpu-baseline", "--no-others",
                   "--no-traffic", "--steps", "4", "--warmup", "1"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=d,
                                   env=dict(os.environ, TMPDIR=d, IPCR_JIT_ASYNC="0", IPCR_BENCH_ONE_WINDOW="1"))
            except Exception:  # noqa: BLE001
                return None
            if r.returncode != 0:
                return None
            acc = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        nm = row.get("Kernel_Name", "")
                        if kernel in nm and not (kernel == "ipcr_filter" and "index" in nm) and row.get("Counter_Name") == counter:
                            acc.append(float(row["Counter_Value"]))
            if not acc:
                return None
            vals[counter] = (sum(acc) / len(acc), len(acc))
    rd = vals["FETCH_SIZE"][0] * 1024.0 * 2.0
    wr = vals["WRITE_SIZE"][0] * 1024.0
    return int(rd + wr), int(min(vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]))


def fasta_to_tsv(ctx, records=8):
    """FASTA file (80-column lines, page cache) -> resident tiles -> scan -> sorted TSV rows, C2 panel."""
    import io
    import tempfile
    import numpy as np
    engine = ctx.engine
    from ipcr_amd import cli
    G = get_genome(ctx, "c2")
    genome = G["g"]
    records = min(records, G["nrec"])
    path = os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), "ipcr_bench_%d.fa" % os.getpid())
    try:
        with open(path, "wb") as fh:
            for r in range(records):
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
        fsize = os.path.getsize(path)
        spec = workload_spec("c2", engine, ctx.workloads)
        eng = engine.New(spec["cfg"])
        cp = eng.CompilePanel(spec["pairs"])
        sc = eng.NewSimulationScratch(cp)
        eng.ScanGenomeCount(genome, cp, sc)                      # kernel build outside the timed stages
        cp.wait_ready()
        best, chunked = None, None
        for _ in range(2):
            t0 = time.perf_counter()
            g = engine.Genome(sum(genome.record_len(r) for r in range(records)) + (1 << 20), max_records=records + 4)
            g.add_fasta(path)
            t1 = time.perf_counter()
            prods = eng.ScanGenome(g, cp, sc)
            t2 = time.perf_counter()
            rows = sorted(((path, p) for p in prods), key=lambda t: cli.product_sort_key(t[0], t[1]))
            out = io.StringIO()
            out.write(cli.TSV_HEADER + "\n")
            for f, p in rows:
                out.write(cli.format_row(f, p) + "\n")
            t3 = time.perf_counter()
            found = {(p.Record, p.Start) for p in prods if p.ExperimentID == "bench_000" and p.Type == "forward"}
            assert all((r, s) in found for (r, s, _) in G["plants"] if r < records), "planted amplicon missing after the FASTA round trip"
            cur = {"file_GB": round(fsize / 1e9, 3), "bases": g.total_bases, "load_s": round(t1 - t0, 4),
                   "scan_ms": round((t2 - t1) * 1e3, 3), "sort_format_ms": round((t3 - t2) * 1e3, 3),
                   "total_s": round(t3 - t0, 4), "gbases_per_s": round(g.total_bases / (t3 - t0) / 1e9, 2),
                   "products": len(prods), "host_threads": len(os.sched_getaffinity(0))}
            if _ == 1:
                # the same file under --chunk-size 4 Mb (overlap = max length + primer length): the resident genome scanned in
                # rolling windows, every window its own ForEachCompiledProduct call (ipcr_scan_genome_chunked: one sweep);
                # window-local products back in record coordinates and deduplicated as the pipeline's collector does
                t4 = time.perf_counter()
                cprods = eng.ScanGenomeChunked(g, cp, sc, 4_000_000, 2020)
                coll = cli.Collector(200_000)
                kept = [q for q in (coll.add(path, p) for p in cprods) if q is not None]
                t5 = time.perf_counter()
                assert {(p.SequenceID, p.Start, p.End, p.ExperimentID, p.Type) for p in kept} == \
                       {(p.SequenceID, p.Start, p.End, p.ExperimentID, p.Type) for p in prods}, "chunked scan of the resident genome differs"
                chunked = {"scan_ms": round((t5 - t4) * 1e3, 3), "products_in_windows": len(cprods), "products": len(kept),
                           "gbases_per_s": round(g.total_bases / ((t1 - t0) + (t5 - t4) + (t3 - t2)) / 1e9, 2)}
            g.close()
            if best is None or cur["total_s"] < best["total_s"]:
                best = cur
        best["chunked_4mb"] = chunked
        sc.close()
        cp.close()
        return best
    finally:
        if os.path.exists(path):
            os.unlink(path)


def cpu_baseline(name, spec, host0, budget_s: float, gpu_products, sample_bases: int = 0, chunk_bases: int = 4_000_000):
    """Reference algorithm restated in C (oracle/: approximate-seed Aho-Corasick scan + verify + join) timed on this box's
    host cores over a bounded sample of the SAME workload: record 0 of the same genome (or its first `sample_bases` bases,
    a record of its own), as many passes queued to ONE worker pool as fill the budget (every thread busy).  One worker per
    rolling chunk like internal/pipeline/pipeline.go:60-125 (4 Mb unless said otherwise); a --circular run (C3) cannot be
    chunked (internal/runutil/runutil.go:46-49), so there one worker scans one whole record, i.e. the pool scans as many
    copies of the sample as it has threads.  Checker/baseline only -- never on the product path.  The sample's product
    count must equal the GPU's for the same bases (gpu_products: the products of record 0, or a count)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ipcr_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    c = spec["cfg"]
    pairs = [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in spec["pairs"]]
    t0 = time.perf_counter()
    panel = O.Panel(O.Config(max_mm=c.MaxMM, terminal_window=c.TerminalWindow, min_len=c.MinLen, max_len=c.MaxLen, hit_cap=c.HitCap,
                             seed_len=c.SeedLen, circular=c.Circular), pairs)
    compile_s = time.perf_counter() - t0
    n = int(host0.shape[0]) if not sample_bases else min(int(sample_bases), int(host0.shape[0]))
    ptr = host0.ctypes.data
    chunk, overlap = (0, 0) if c.Circular else (chunk_bases, 2000)
    nch = 1 if c.Circular else max(1, -(-max(n - overlap, 1) // (chunk - overlap)))
    passes = max(1, -(-cores // nch))                           # calibration round: every thread gets a chunk
    t0 = time.perf_counter()
    nprod, busy, nch = panel.baseline_scan_pool(ptr, n, chunk, overlap, cores, passes)
    el = time.perf_counter() - t0
    rate = n * passes / el
    if el < budget_s * 0.6:                                     # fill the budget with ONE longer pool run
        passes2 = max(passes, int(passes * (budget_s / max(el, 1e-3)) * 0.8))
        t0 = time.perf_counter()
        nprod, busy, nch = panel.baseline_scan_pool(ptr, n, chunk, overlap, cores, passes2)
        el = time.perf_counter() - t0
        rate, passes = n * passes2 / el, passes2
    panel.close()
    gpu_n = gpu_products if isinstance(gpu_products, int) else len([p for p in gpu_products if p.Record == 0])
    assert nprod == gpu_n, f"CPU baseline found {nprod} products in its sample, the GPU path {gpu_n}"
    how = ("whole records (a --circular run is not chunked), one worker per record" if c.Circular
           else "%d chunks of %.1f Mb, overlap 2000, per pass" % (nch, chunk / 1e6))
    what = "record 0 (%d bases)" % n if not sample_bases else "the first %d bases of record 0, as a record of its own," % n
    return {
        "value": round(rate / 1e9, 4),
        "unit": "Gbases/s",
        "cores": cores,
        "threads_busy": int(busy),
        "kind": "port",
        "workload": name,
        "panel_compile_s": round(compile_s, 3),
        "sample": "%s of the same genome, %d passes queued to one pool of %d threads (%s = %d scans in all, "
                  "%.1f s); C restatement of the reference's seeded AC scan + verify + join (not the Go binary)"
                  % (what, passes, cores, how, nch * passes, el),
        "products_in_sample": int(nprod),
    }


def cpu_baseline_bounded(ctx, name, seconds: float = 3.0):
    """cpu_baseline for a workload that is not the main line (other_workloads.<name>.cpu_baseline), bounded to a few
    seconds: a prefix of record 0 sized so that one pool run of every host thread fits -- C3 (k = 3, whole records per
    worker) 8 Mb per thread; C4 (the 1024-row panel: a DRAM-resident automaton, ~0.1 Mb/s per thread) 4 Mb in rolling
    chunks of 0.5 Mb.  The GPU path scans the same bytes (ipcr_scan_chunk) for the product count the sample must match."""
    engine = ctx.engine
    spec = workload_spec(name, engine, ctx.workloads)
    G = get_genome(ctx, spec["genome"])
    host0 = G["host0"]
    if host0 is None:
        return None
    n, chunk = (8_000_000, 0) if name == "c3" else (4_000_000, 500_000)
    n = min(n, int(host0.shape[0]))
    sample = host0[:n].tobytes()
    gpu_n = len(engine.New(spec["cfg"]).SimulateBatch("sample", sample, spec["pairs"]))
    return cpu_baseline(name, spec, host0, seconds, gpu_n, sample_bases=n, chunk_bases=chunk or 4_000_000)


# ----------------------------------------------------------------------------------------------- main
def main() -> None:
    # HIP maps streams onto a few hardware queues (4 by default), in creation order.  This job has a dozen (three
    # scratches, the genome, torch, RCCL): when the sweep lane shares a queue with the collective's stream, every
    # all-gather is serialised between two sweeps (+25 us per step measured).  One queue per stream instead.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c2n", "c3", "c4", "c4n", "c5"])
    # one C2 step is ~0.2 ms: the default region (about half a second) is long enough for the clocks to settle
    ap.add_argument("--steps", type=int, default=None, help="timed passes (default 2000; 40 for c4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed passes before them (default 200; 5 for c4)")
    ap.add_argument("--records", type=int, default=RECORDS)
    ap.add_argument("--record-len", type=int, default=RECORD_LEN)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip config.other_workloads")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo to rehearse)")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every pass before the next one is enqueued")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc children (quote profiles/ instead)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 40 if args.workload in ("c4", "c4n") else 2000
    if args.warmup is None:
        args.warmup = 5 if args.workload in ("c4", "c4n") else 200

    cmd = launch_plan(args.gpus, os.environ, sys.argv[1:])
    if cmd is not None:   # launcher only: no torch import, no HIP call in this process
        import subprocess
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ)))

    chunk_rates, c5_chunk, all_dev_rates, traffic, chunk_panel = None, None, None, None, None
    single = "RANK" not in os.environ and args.gpus <= 1 and not os.environ.get("IPCR_EXCHANGE_SELFTEST")
    if not args.no_others and single:
        chunk_rates = scan_chunk_rates(args)      # child process, before anything here has initialised HIP
        c5_chunk = scan_chunk_rates(args, probe=True)
        # the drop-in call under C4's panel (1024 rows: seed-index kernel), 16 workers: clean chunks and chunks that hold N
        chunk_panel = {"panel_rows": 1024, "workers": 16, "chunk_bases": 4_000_000}
        for key, wn in (("clean", False), ("with_n", True)):
            cr = scan_chunk_rates(args, with_n=wn, panel_rows=1024, workers=(16,))
            chunk_panel["gbases_per_s_" + key] = cr["gbases_per_s_16_workers"]
            chunk_panel["calls_unwindowed_" + key] = cr["calls_unwindowed_16_workers"]
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if not args.no_traffic and not under_profiler and single and args.records == RECORDS and args.record_len == RECORD_LEN:
        traffic = measure_traffic(args.workload, "ipcr_index_filter" if args.workload in ("c4", "c4n") else "ipcr_filter")
    if not args.no_others and os.environ.get("RANK") == "0" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        all_dev_rates = scan_chunk_all_devices(int(os.environ["WORLD_SIZE"]), args)

    import glob
    import torch
    from ipcr_amd import _lib, dist, engine, workloads, primer

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the ipcr_amd scan path has no CPU fallback")
    if os.environ.get("IPCR_BENCH_ONE_DEVICE"):  # rehearsal: several ranks share GPU 0 (gloo collectives)
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local, backend = dist.init_process_group(args.backend)
    torch.cuda.set_device(local)
    _lib.check(_lib.lib().ipcr_set_device(local))
    import torch.distributed as tdist

    ctx = Ctx()
    ctx.torch, ctx.engine, ctx.workloads, ctx.dist, ctx.tdist, ctx._lib = torch, engine, workloads, dist, tdist, _lib
    ctx.args, ctx.rank, ctx.world, ctx.local, ctx.backend = args, rank, world, local, backend
    ctx.dev = torch.device("cuda", local)
    ctx.cdev = ctx.dev if backend == "nccl" else torch.device("cpu")  # where collective payloads live
    ctx.multi = world > 1 or bool(os.environ.get("IPCR_EXCHANGE_SELFTEST"))  # selftest: one rank runs the RCCL exchange too
    ctx.revcomp = primer.RevComp
    ctx.genomes = {}
    ctx.want_cpu = (not args.no_cpu_baseline) and rank == 0 and world == 1 and args.workload in ("c2", "c3", "c4")
    ctx.want_cpu_others = (not args.no_cpu_baseline) and (not args.no_others) and rank == 0 and world == 1 and not ctx.multi
    ctx.compile_done = set()

    res = run_workload(ctx, args.workload, args.steps, args.warmup)

    others = {}
    if not args.no_others:
        if world == 1 and not ctx.multi:
            for nm, st, wu in (("c2n", 400, 100), ("c3", 400, 100), ("c4", 30, 5), ("c4n", 20, 3), ("c5", 300, 50)):
                if nm == args.workload:
                    continue
                r = run_workload(ctx, nm, st, wu)
                opmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_pmc.json" % nm)))
                rf = roofline_of(r, opmc[-1] if opmc else None)
                others[nm] = {"workload": r["text"], "ms_per_step": round(r["ms_per_step"], 4), "gbases_per_s": round(r["value"], 1),
                              "sweep_ms": round(r["filter_ms"], 4), "kernel": r["kernel"], "roofline_frac": rf["frac"],
                              "steps": r["steps"], "warmup_actual": r["warmup_actual"], "products_per_step": r["nprod"],
                              "hits_per_step": r["hits"], "planted_amplicons_verified": r["plants"],
                              "step_breakdown_ms": r["breakdown"], "panel_compile": r["compile_times"]}
                if r["probe_ms"] is not None:
                    others[nm]["probe_rescan_ms"] = round(r["probe_ms"], 4)
                if "limiter" in rf:
                    others[nm]["limiter"] = rf["limiter"]
                if nm in ("c3", "c4") and ctx.want_cpu_others:   # the driver's record carries every workload's CPU baseline
                    others[nm]["cpu_baseline"] = cpu_baseline_bounded(ctx, nm)
                r["prods"] = None
            if chunk_rates is not None:
                others["scan_chunk"] = chunk_rates
            if chunk_panel is not None:
                others["scan_chunk_c4_panel"] = chunk_panel
            if c5_chunk is not None:    # ipcr-probe over ipcr_scan_chunk + ipcr_probe_scratch_products, 16 workers; ipcr_probe_best_hit latency
                others["c5_chunk"] = c5_chunk
            others["fasta_to_tsv"] = fasta_to_tsv(ctx)
        elif args.workload != "c4":   # several GPUs: the scaling target north_star names rides along
            r = run_workload(ctx, "c4", 30, 5)
            others["c4"] = {"workload": r["text"], "ms_per_step": round(r["ms_per_step"], 4), "gbases_per_s": round(r["value"], 1),
                            "sweep_ms_rank0": round(r["filter_ms"], 4), "kernel": r["kernel"], "n_gpus": world, "scaling": "weak",
                            "steps": r["steps"], "warmup_actual": r["warmup_actual"], "products_per_step": r["nprod"],
                            "device_path": r["device_path"], "exchange_redone": r["exchange_redone"],
                            "native_exchange": r["native_exchange"], "native_exchange_verified": r["native_exchange_verified"]}
            r["prods"] = None
        if all_dev_rates is not None:
            others["scan_chunk_one_process_all_devices"] = all_dev_rates

    genome_bases = res["bases_per_gpu"]
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_pmc.json" % ("filter" if args.workload == "c2" else args.workload))))
    out = {
        "metric": "genome Gbases scanned/sec (k=%d)" % (3 if args.workload == "c3" else 2),
        "value": round(res["value"], 2),
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "warmup_actual": res["warmup_actual"],
        "ms_per_step": round(res["ms_per_step"], 4),
        "windows": res["windows"],                      # timed windows of `steps` passes each; ms_per_step / value are the median window's
        "window_ms_min": round(res["window_ms_min"], 4),
        "window_ms_max": round(res["window_ms_max"], 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": "%s, synthetic %.2f Gb genome per GPU (%d x %d b LCG records, %d planted amplicons verified)"
                        % (res["text"], genome_bases / 1e9, res["nrec"], args.record_len, res["plants"]),
            "input": "2-bit + invalid-bit tiles resident in HBM (0.375 B/base); pack kernel timed separately",
            "products_per_step": res["nprod"],
            "hits_per_step_rank0": res["hits"],
            "filter_candidates_rank0": res["candidates"],
            "filter_kernel": {1: "panel-specialised (hiprtc)", 2: "table-driven", 3: "seed-index (hiprtc)"}.get(res["kernel_kind"], "?"),
            "n_patterns": res["n_patterns"],
            "pipelining": "off" if args.no_pipeline else
                          "pass i+1's sweep is queued behind pass i's on one in-order stream while the host waits for, sorts "
                          "and joins pass i; the specialised sweep verifies its own survivors and its last wave publishes "
                          "counters + hits to pinned memory (no verify kernel, no copy operation)",
            "pack_ms_per_genome": round(res["pack_ms"], 3),
            "gbases_per_s_incl_pack": round(genome_bases * world / ((res["pack_ms"] + res["ms_per_step"]) * 1e-3) / 1e9, 1),
            "step_breakdown_ms_rank0": res["breakdown"],
            "rccl_ranks": world if ctx.multi else 0,
            "backend": backend if ctx.multi else None,
            "device_path": res["device_path"],
            "native_exchange": res["native_exchange"],   # ncclAllGather inside libipcr_hip.so (csrc/exchange.cpp), not torch.distributed
            "native_exchange_verified": res["native_exchange_verified"],   # ... and its first result equal to torch.distributed's on every rank
            "exchange_redone": res["exchange_redone"],
            "panel_compile": res["compile_times"],
            "parallelism": ("1 genome per GPU (weak scaling); one all-gatherv of hit records per step (%s, %s; %d exchanges redone "
                            "after an overflow); every rank joins its own records inside the step, the gathered records are "
                            "joined whole on rank 0 once after the timed region as a check"
                            % (backend, "out of the device hit buffer" if res["device_path"] else "host copy of the hits",
                               res["exchange_redone"])) if ctx.multi else "single GPU",
            "other_workloads": others,
        },
        "roofline": roofline_of(res, pmc[-1] if pmc else None),
    }
    if traffic is not None:     # measured by this run (measure_traffic), not quoted
        out["roofline"]["traffic"] = traffic[0]
        out["roofline"]["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE children of this command "
                                             "(separate passes, %d launches averaged; FETCH_SIZE doubled, MI355X_MICROARCH.md HBM)" % traffic[1])
        out["roofline"]["traffic_over_algorithmic"] = round(traffic[0] / out["roofline"]["algorithmic_bytes_per_launch"], 4)
    if res["probe_ms"] is not None:
        out["config"]["probe_rescan_ms"] = round(res["probe_ms"], 4)
    if ctx.want_cpu:
        spec = workload_spec(args.workload, engine, workloads)
        out["cpu_baseline"] = cpu_baseline(args.workload, spec, ctx.genomes[spec["genome"]]["host0"], args.cpu_seconds, res["prods"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if ctx.multi and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
