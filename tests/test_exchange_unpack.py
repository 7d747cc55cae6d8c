"""The host side of the all-gatherv of hit records (csrc/exchange.cpp: ipcr_exchange_end) for world > 1, on the CPU.

On the GPU boxes of the build pool a communicator never has more than one rank, so the multi-rank logic -- per-rank
counts out of the gathered headers, the lock-step overflow decision, where rank r's records start in the gathered
buffer, record rebasing -- is driven here through ipcr_exchange_unpack, the same code as a pure function over a
gathered buffer in host memory (ipcr_exchange_end calls plan_gather / records_offset / rebase_records, and so does
it).  Expected results are built independently with numpy."""
import ctypes as C

import numpy as np
import pytest

from ipcr_amd import _lib
from ipcr_amd.dist import HIT_DTYPE


def make_block(cap, hits, cset=0, reported=None):
    """one rank's device hit block as ipcr_scratch_device_hits lays it out: 64-byte header (two counter sets of four
    uint64: queue words, hits, candidates, fullest segment; the last scan's set is the non-zero one) + cap slots"""
    blk = np.zeros(64 + cap * 32, dtype=np.uint8)
    hdr = blk[:64].view(np.uint64)
    n = len(hits) if reported is None else reported
    hdr[4 * cset + 0] = 7            # queue words: must not be mistaken for the hit count
    hdr[4 * cset + 1] = n
    hdr[4 * cset + 2] = 3 * n + 1    # candidate windows
    m = min(len(hits), cap)
    if m:
        blk[64:64 + m * 32] = hits[:m].view(np.uint8)
    blk[64 + m * 32:] = 0xEE         # stale bytes behind the valid prefix must never be read as records
    return blk


def rand_hits(rng, n, nrec):
    h = np.zeros(n, dtype=HIT_DTYPE)
    h["pos"] = rng.integers(0, 1 << 33, n)
    h["record"] = rng.integers(0, max(nrec, 1), n)
    h["pattern"] = rng.integers(0, 4096, n) | (rng.integers(0, 2, n).astype(np.uint32) << 31)
    h["mm0"] = rng.integers(0, 1 << 20, n)
    h["mm1"] = rng.integers(0, 4, n)
    return h


def unpack(gathered, world, cap, rec_counts, same, out_cap=None):
    L = _lib.lib()
    total_cap = out_cap if out_cap is not None else world * cap
    out = np.zeros(max(total_cap, 1), dtype=HIT_DTYPE)
    starts = (C.c_uint64 * (world + 1))()
    offs = (C.c_uint32 * (world + 1))()
    need = C.c_uint64(0)
    rc = (C.c_uint32 * world)(*rec_counts) if rec_counts is not None else None
    st = L.ipcr_exchange_unpack(gathered.ctypes.data, world, cap, rc, int(same), out.ctypes.data, total_cap, starts, offs, C.byref(need))
    return st, out, list(starts), list(offs), need.value


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("same", [False, True])
def test_unpack_uneven_counts(world, same):
    rng = np.random.default_rng(1000 + world * 2 + same)
    cap = 37
    rec_counts = [int(rng.integers(0, 30)) for _ in range(world)]
    counts = [int(rng.integers(0, cap + 1)) for _ in range(world)]
    counts[rng.integers(0, world)] = cap                     # one rank exactly full
    if world > 2:
        counts[1] = 0                                        # a rank without hits
    per_rank = [rand_hits(rng, counts[r], rec_counts[r] if not same else 11) for r in range(world)]
    gathered = np.concatenate([make_block(cap, per_rank[r], cset=r & 1) for r in range(world)])
    st, out, starts, offs, need = unpack(gathered, world, cap, rec_counts, same)
    assert st == _lib.OK and need == max(counts)
    assert starts == [sum(counts[:r]) for r in range(world + 1)]
    want_offs = [0] * (world + 1) if same else [sum(rec_counts[:r]) for r in range(world + 1)]
    assert offs == want_offs
    for r in range(world):
        part = out[starts[r]:starts[r + 1]]
        want = per_rank[r].copy()
        want["record"] += np.uint32(want_offs[r])
        assert part.tobytes() == want.tobytes(), (world, same, r)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_unpack_overflow_is_the_same_decision_on_every_rank(world):
    """A rank with more hits than the capacity still enters the collective: header with the TRUE count, the first cap
    records.  Every rank reads the same headers -> the same status and the same `need` -> all repeat the exchange with
    the doubled capacity, which then unpacks completely."""
    rng = np.random.default_rng(77 + world)
    cap = 16
    rec_counts = [5] * world
    true_counts = [int(rng.integers(0, cap)) for _ in range(world)]
    over = world - 1
    true_counts[over] = 3 * cap + 5
    per_rank = [rand_hits(rng, true_counts[r], 5) for r in range(world)]
    gathered = np.concatenate([make_block(cap, per_rank[r]) for r in range(world)])
    results = [unpack(gathered.copy(), world, cap, rec_counts, False) for _ in range(world)]   # what each rank computes
    assert all(st == _lib.ERR_CAPACITY and need == true_counts[over] for st, _, _, _, need in results)
    assert all(r[2] == results[0][2] for r in results)
    cap2 = cap
    while cap2 < results[0][4]:
        cap2 *= 2                                                                                 # exchange.cpp: the redo's capacity
    gathered2 = np.concatenate([make_block(cap2, per_rank[r]) for r in range(world)])
    st, out, starts, offs, need = unpack(gathered2, world, cap2, rec_counts, False)
    assert st == _lib.OK and starts[-1] == sum(true_counts)
    for r in range(world):
        want = per_rank[r].copy()
        want["record"] += np.uint32(5 * r)
        assert out[starts[r]:starts[r + 1]].tobytes() == want.tobytes()


def test_unpack_argument_checks_and_output_bound():
    rng = np.random.default_rng(5)
    cap, world = 8, 3
    per_rank = [rand_hits(rng, 8, 2) for _ in range(world)]
    gathered = np.concatenate([make_block(cap, h) for h in per_rank])
    st, *_ = unpack(gathered, world, cap, [2, 2, 2], False, out_cap=23)      # 24 records do not fit 23 slots
    assert st == _lib.ERR_CAPACITY
    L = _lib.lib()
    assert L.ipcr_exchange_unpack(None, world, cap, None, 1, None, 0, None, None, None) == _lib.ERR_INVALID
    assert L.ipcr_exchange_unpack(gathered.ctypes.data, 0, cap, None, 1, None, 0, None, None, None) == _lib.ERR_INVALID
    assert L.ipcr_exchange_unpack(gathered.ctypes.data, world, cap, None, 0, None, 0, None, None, None) == _lib.ERR_INVALID  # record counts needed
    # counts only (no output buffer): a host sizing its buffers
    need = C.c_uint64()
    starts = (C.c_uint64 * (world + 1))()
    st = L.ipcr_exchange_unpack(gathered.ctypes.data, world, cap, None, 1, None, 0, starts, None, C.byref(need))
    assert st == _lib.ERR_CAPACITY and need.value == 8 and list(starts) == [0, 8, 16, 24]


def test_unpacked_hits_join_like_one_rank(tmp_path):
    """genome-parallel job on the CPU: two 'ranks' hold different records; their hit lists, gathered and unpacked, joined
    with ipcr_join_hits give the products of the whole job in job-global record order (the oracle finds the same)."""
    import ipcr_oracle as O
    from ipcr_amd import engine, primer
    rng = np.random.default_rng(9)
    fwd, rev = "ACGTTGCATGCAAGCT", "GGCCTTAAGGCCATAT"
    pairs = [primer.Pair("p", fwd, rev, 0, 0)]
    cfg = engine.Config(MaxMM=1, TerminalWindow=3, MaxLen=1000, HitCap=100, SeedLen=12)
    ocfg = O.Config(max_mm=1, terminal_window=3, max_len=1000, hit_cap=100, seed_len=12)
    opairs = [O.Pair("p", fwd, rev, 0, 0)]
    cp = engine.New(cfg).CompilePanel(pairs)
    recs = []
    for r in range(5):
        s = bytearray(O.bench_dna(3000, 900 + r))
        s[500:500 + len(fwd)] = fwd.encode()
        rc = O.revcomp(rev)
        s[500 + 300 - len(rc):500 + 300] = rc
        recs.append(bytes(s))
    shards = [recs[:2], recs[2:]]                       # rank 0: records 0-1, rank 1: records 2-4
    blocks, rec_counts = [], []
    pat = {(i, w): cp.slot_pattern(i, w, 0) for i in range(len(pairs)) for w in "ABab"}
    for shard in shards:
        hits = []
        for li, s in enumerate(shard):
            for w in "ABab":
                patseq, left, _tw, _, _ = cp.pattern_info(pat[(0, w)])
                for m in O.find_matches(s, patseq, 1, 0, 0):
                    idx = m.idx
                    if left and any(j < 3 for j in idx):
                        continue
                    if not left and any(j >= len(patseq) - 3 for j in idx):
                        continue
                    hits.append((m.pos, li, pat[(0, w)], sum(1 << j for j in idx), 0))
        arr = np.array(hits, dtype=HIT_DTYPE)
        rng.shuffle(arr)                                # device append order is no order
        blocks.append(make_block(64, arr))
        rec_counts.append(len(shard))
    st, out, starts, offs, _ = unpack(np.concatenate(blocks), 2, 64, rec_counts, False)
    assert st == _lib.OK and offs == [0, 2, 5]
    sc = engine.SimulationScratch(cp, host_only=True)
    got = engine.New(cfg).JoinHits(cp, sc, out[:starts[-1]], [len(s) for s in recs], [0] * 5)
    want = []
    for r, s in enumerate(recs):
        want += [(str(r),) + w.sig() for w in O.simulate_batch(ocfg, s, opairs)]
    assert [(p.SequenceID,) + p.sig() for p in got] == want and len(want) >= 5
    sc.close()
    cp.close()
