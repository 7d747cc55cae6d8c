"""Drop-in entry point under the reference's worker model (internal/pipeline/pipeline.go:60-125): W worker threads, one
scratch each, share one compiled panel and pull rolling chunks of one record from a queue; every chunk goes through
ipcr_scan_chunk as host ASCII.  Prints the aggregate PCIe-inclusive rate next to the raw pinned H2D rate (dev tool;
bench.py reports the same numbers in config.other_workloads).

    python tools/chunk_workers_probe.py [record_bases] [chunk_bases] [workers ...]
"""
import os, sys, time, threading, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipcr_amd import engine, workloads


def h2d_rate(nbytes=256 << 20, reps=5):
    """what the link gives: pinned host -> device copies, best of reps"""
    h = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    return best


def run(seq, chunk, overlap, workers, eng, cp, passes=3):
    n = len(seq)
    starts = list(range(0, n, chunk - overlap)) if n > chunk else [0]
    view = memoryview(seq)
    scs = [eng.NewSimulationScratch(cp) for _ in range(workers)]
    for sc in scs:                                   # kernel build + buffer sizing outside the timed region
        eng.SimulateCompiledWithScratch("w", bytes(view[:chunk]), cp, sc)
    chunks = [bytes(view[s:s + chunk]) for s in starts]   # the jobs own private copies, as fasta/path_ctx.go:117 makes them
    best = 0.0
    nprod = 0
    reps = max(1, -(-16 * workers // len(chunks)))      # >= 16 chunks per worker in a timed pass
    for _ in range(passes):
        q = queue.Queue()
        for rep in range(reps):                         # enough jobs per worker that thread start-up does not show
            for i, c in enumerate(chunks):
                q.put((i, c))
        counts = [0] * workers

        def work(w):
            sc = scs[w]
            while True:
                try:
                    i, c = q.get_nowait()
                except queue.Empty:
                    return
                counts[w] += len(eng.SimulateCompiledWithScratch("chr:%d-%d" % (starts[i], starts[i] + len(c)), c, cp, sc))

        ths = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        best = max(best, reps * sum(len(c) for c in chunks) / dt / 1e9)
        nprod = sum(counts) // reps
    for sc in scs:
        sc.close()
    return best, nprod, len(chunks)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
    workers = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8, 16]
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    engine.lcg_fill_device(buf.data_ptr(), n, 0x5eed1234)
    seq = buf.cpu().numpy().tobytes()
    del buf
    eng = engine.New(engine.Config(MaxMM=2, TerminalWindow=5, MaxLen=2000, HitCap=10000, SeedLen=12))
    cp = eng.CompilePanel(workloads.c2_pairs())
    print(f"pinned H2D: {h2d_rate():.1f} GB/s", flush=True)
    for w in workers:
        rate, nprod, nch = run(seq, chunk, 2000, w, eng, cp)
        print(f"scan_chunk: {w:2d} workers x {chunk/1e6:.0f} Mb chunks ({nch} chunks of a {n/1e6:.0f} Mb record): {rate:.2f} Gbases/s aggregate, {nprod} products", flush=True)
    rate, nprod, _ = run(seq, n, 0, 1, eng, cp)
    print(f"scan_chunk: whole {n/1e6:.0f} Mb record, 1 worker: {rate:.2f} Gbases/s", flush=True)


if __name__ == "__main__":
    main()
