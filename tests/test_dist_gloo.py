"""N>1 path on CPU: world_size-2 gloo job running the shard -> all-gatherv of hit records -> join
pipeline of ipcr_amd.dist (hits synthesised on the CPU, as in test_host_logic)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, random
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch.distributed as tdist
import ipcr_oracle as O
from ipcr_amd import dist, engine, primer
from test_host_logic import synth_hits, rand_seq, plant

rank, world, local, backend = dist.init_process_group("gloo")
assert world == 2 and backend == "gloo"
rng = random.Random(5)  # same stream on both ranks: both build the same 5 records
pair = primer.Pair("p", "ACGTTGCATGCAAGCT", "GGCCTTAAGGCCATAT")
pairs = primer.AddSelfPairs([pair])
cfg = engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=300, HitCap=10000, SeedLen=12)
seqs = []
for r in range(5):
    s = rand_seq(rng, 4000, junk=(r == 3))
    for _ in range(4):
        a = rng.randrange(0, 3500)
        plant(rng, s, pair.Forward, a, rng.choice([0, 1, 2]))
        plant(rng, s, O.revcomp(pair.Reverse).decode(), a + rng.randint(40, 250), rng.choice([0, 1]))
    seqs.append("".join(s).encode())
mine = dist.shard_range(len(seqs), rank, world)          # records sharded over ranks
assert list(mine) == ([0, 1, 2] if rank == 0 else [3, 4])
eng = engine.New(cfg)
cp = eng.CompilePanel(pairs)
reset = [any(ch not in b"ACGTacgt" for ch in seqs[g]) for g in mine]
mode = 1 if any(reset) else 0
local_hits = np.concatenate([synth_hits(cp, seqs[g], cfg.MaxMM, i, mode) for i, g in enumerate(mine)])
lens = [len(seqs[g]) for g in mine]
flags = [(1 if reset[i] else 0) | (2 if mode else 0) for i in range(len(lens))]
all_hits, offsets = dist.allgather_hits(local_hits, len(lens))      # the all-gatherv
xchg = dist.HitExchanger(cap_hits=4)                       # tiny capacity: forces the agreed regrow round
x_hits, x_ranges, x_offsets = xchg.allgather(local_hits, len(lens))
assert x_offsets == offsets and np.array_equal(x_hits, all_hits) and xchg.cap >= max(b - a for a, b in x_ranges)
a, b = x_ranges[rank]
assert np.array_equal(x_hits[a:b]["pos"], local_hits["pos"])
x2, _, _ = xchg.allgather(local_hits, len(lens))          # steady state: one collective
assert np.array_equal(x2, all_hits)
assert xchg.agree_on_device_path(None) is False             # CPU job: every rank settles on the host copy (one all-reduce)
work = xchg.start(local_hits, len(lens))                   # overlapped form: enqueue, join own records, wait
xchg.finish(work)
x3, r3, o3 = xchg.gathered()
assert np.array_equal(x3, all_hits) and r3 == x_ranges and o3 == offsets
# overflow on ONE rank of two in the overlapped form: nobody raises before the collective, nobody hangs; every rank
# sees the overflow in the gathered headers and they regrow + redo together (ADVICE r1: lock-step overflow handling)
lens_by_rank = [b - a for a, b in x_ranges]
small = dist.HitExchanger(cap_hits=min(lens_by_rank) if min(lens_by_rank) < max(lens_by_rank) else max(lens_by_rank) - 1)
assert small.cap < max(lens_by_rank)
w = small.start(local_hits, len(lens))
small.finish(w)
x4, r4, o4 = small.gathered()
assert small.redone == 1 and small.cap >= max(lens_by_rank)
assert np.array_equal(x4, all_hits) and r4 == x_ranges and o4 == offsets
w = small.start(local_hits, len(lens)); small.finish(w)       # steady state again: fits, one collective
assert small.redone == 1 and np.array_equal(small.gathered()[0], all_hits)
all_lens, all_flags = dist.allgather_record_meta(lens, flags)
assert offsets == [0, 3] and all_lens == [4000] * 5 and len(all_flags) == 5
sc = engine.SimulationScratch(cp, host_only=True)
got = eng.JoinHits(cp, sc, all_hits, all_lens, all_flags, ["rec%d" % r for r in range(5)])
op = O.Panel(O.Config(max_mm=2, terminal_window=3, max_len=300, hit_cap=10000, seed_len=12),
             [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs])
want = []
for r, s in enumerate(seqs):
    want += [("rec%d" % r,) + w.sig() for w in op.scan(s)]
assert [(g.SequenceID,) + g.sig() for g in got] == want and len(want) >= 10
# join partitioned by record: each rank joins its own slice of the gathered hits
mine_got = eng.JoinHits(cp, sc, x_hits[a:b], all_lens, all_flags, ["rec%d" % r for r in range(5)])
mine_want = [w for w in want if int(w[0][3:]) in mine]
assert [(g.SequenceID,) + g.sig() for g in mine_got] == mine_want
tdist.barrier()
print("RANK_OK", rank, len(got))
'''


def test_two_rank_gloo_allgather_and_join(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"RANK_OK {rank}" in out, out


SHARD_WORKER = r'''
import os, sys, random
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch.distributed as tdist
import ipcr_oracle as O
from ipcr_amd import dist, engine, primer
from test_host_logic import synth_hits, rand_seq, plant

rank, world, local, backend = dist.init_process_group("gloo")
assert world == 2
rng = random.Random(31)   # same stream on both ranks: same panel, same records
rows = [primer.Pair("r%d" % i, "".join(rng.choice("ACGT") for _ in range(19)), "".join(rng.choice("ACGT") for _ in range(21)), 0, 0) for i in range(6)]
pairs = primer.AddSelfPairsUnique(rows)
cfg = engine.Config(MaxMM=2, TerminalWindow=3, MaxLen=500, HitCap=10000, SeedLen=12)
seqs = []
for r in range(4):
    s = rand_seq(rng, 5000, junk=(r == 2))
    for i in range(6):
        a = 100 + i * 780
        plant(rng, s, rows[i].Forward, a, rng.choice([0, 1, 2]))
        plant(rng, s, O.revcomp(rows[i].Reverse).decode(), a + rng.randint(60, 300), rng.choice([0, 1]))
    seqs.append("".join(s).encode())
eng = engine.New(cfg)
full = eng.CompilePanel(pairs)
mine = dist.pattern_shard(eng.CompilePanel(pairs))          # this rank's slice of the distinct-pattern list
assert len(mine.scanned_patterns(0)) * 2 in (len(full.scanned_patterns(0)), len(full.scanned_patterns(0)) + 1, len(full.scanned_patterns(0)) - 1)
reset = [any(ch not in b"ACGTacgt" for ch in s) for s in seqs]
mode = 1 if any(reset) else 0
local_hits = np.concatenate([synth_hits(mine, s, cfg.MaxMM, i, mode) for i, s in enumerate(seqs)])   # EVERY record, my patterns
lens = [len(s) for s in seqs]
flags = [(1 if reset[i] else 0) | (2 if mode else 0) for i in range(len(seqs))]
all_hits, offsets = dist.allgather_hits(local_hits, len(seqs), same_records=True)
assert offsets == [0, 0] and len(all_hits) > len(local_hits)
x = dist.HitExchanger(cap_hits=8, same_records=True)          # the persistent form, with an agreed regrow on the way
x_hits, _, x_off = x.allgather(local_hits, len(seqs))
assert x_off == [0, 0] and np.array_equal(x_hits, all_hits)
w = x.start(local_hits, len(seqs)); x.finish(w)
assert np.array_equal(x.gathered()[0], all_hits)
sc = engine.SimulationScratch(full, host_only=True)
got = eng.JoinHits(full, sc, all_hits, lens, flags, ["rec%d" % r for r in range(len(seqs))])
op = O.Panel(O.Config(max_mm=2, terminal_window=3, max_len=500, hit_cap=10000, seed_len=12),
             [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in pairs])
want = []
for r, s in enumerate(seqs):
    want += [("rec%d" % r,) + w_.sig() for w_ in op.scan(s)]
assert [(g.SequenceID,) + g.sig() for g in got] == want and len(want) >= 12
whole = np.concatenate([synth_hits(full, s, cfg.MaxMM, i, mode) for i, s in enumerate(seqs)])   # the single-rank scan
single = eng.JoinHits(full, sc, whole, lens, flags, ["rec%d" % r for r in range(len(seqs))])
assert [g.sig() for g in single] == [g.sig() for g in got]
tdist.barrier()
print("SHARD_OK", rank, len(got))
'''


def test_two_rank_gloo_pattern_shards(tmp_path):
    """pattern-axis sharding (north_star "primer x genome tiles"): two ranks scan the SAME records with
    halves of the distinct-pattern list, all-gather the hits (no record rebase) and join with the full panel:
    the product list equals the single-rank scan's and the oracle's"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"SHARD_OK {rank}" in out, out


NATIVE_WORKER = r'''
import os, sys, random, ctypes as C
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch, torch.distributed as tdist
from ipcr_amd import _lib, dist

rank, world, local, backend = dist.init_process_group("gloo")
assert world == 2
REAL = _lib.lib()

class FakeScratch:                      # what ipcr_exchange_begin reads of a scratch: its device hit block
    def __init__(self, hits): self.hits = hits; self._h = id(self)
SCRATCHES = {{}}

class FakeExchangeLib:
    """ipcr_exchange_* with the collective carried by gloo and everything after it by the REAL ipcr_exchange_unpack:
    the entry points HitExchanger's native caller uses, with the semantics of csrc/exchange.cpp (true count in the header,
    first cap records, lock-step redo inside end, two slots)."""
    def __init__(self): self.x = None
    def ipcr_last_error(self): return REAL.ipcr_last_error()
    def ipcr_exchange_available(self, dev): return 1
    def ipcr_exchange_unique_id(self, buf): return 0
    def ipcr_exchange_create(self, uid, world, rank, dev, cap, same, out):
        self.x = dict(world=world, rank=rank, cap=max(int(cap), 1), same=int(same), counts=[0] * world, pend={{}}, next=0, redone=0, keep=None)
        out._obj.value = 1
        return 0
    def ipcr_exchange_destroy(self, x): self.x = None
    def ipcr_exchange_set_record_counts(self, x, arr): self.x["counts"] = [int(arr[r]) for r in range(self.x["world"])]; return 0
    def ipcr_exchange_reserve(self, x, cap): self.x["cap"] = max(self.x["cap"], int(cap)); return 0
    def ipcr_exchange_capacity(self, x): return self.x["cap"]
    def ipcr_exchange_redone(self, x): return self.x["redone"]
    def _gather(self, hits, cap):
        blk = np.zeros(64 + cap * 32, dtype=np.uint8)
        blk[:64].view(np.uint64)[1] = len(hits)
        m = min(len(hits), cap)
        if m: blk[64:64 + m * 32] = hits[:m].view(np.uint8).reshape(-1)
        recv = torch.empty(self.x["world"] * len(blk), dtype=torch.uint8)
        tdist.all_gather_into_tensor(recv, torch.from_numpy(blk))
        return recv.numpy()
    def ipcr_exchange_begin(self, x, scratch_h, ticket):
        slot = self.x["next"]
        if slot in self.x["pend"]: return 1
        hits = SCRATCHES[scratch_h].hits
        self.x["pend"][slot] = (hits, self.x["cap"], self._gather(hits, self.x["cap"]))
        self.x["next"] = (slot + 1) % 2
        ticket._obj.value = slot
        return 0
    def ipcr_exchange_end(self, x, t, hits_p, n_p, starts_p, offs_p):
        hits, cap, gathered = self.x["pend"].pop(t)
        w = self.x["world"]
        for _ in range(24):
            out = np.zeros(max(w * cap, 1), dtype=dist.HIT_DTYPE)
            starts, offs, need = (C.c_uint64 * (w + 1))(), (C.c_uint32 * (w + 1))(), C.c_uint64()
            st = REAL.ipcr_exchange_unpack(gathered.ctypes.data, w, cap, (C.c_uint32 * w)(*self.x["counts"]), self.x["same"],
                                           out.ctypes.data, w * cap, starts, offs, C.byref(need))
            if st == _lib.ERR_CAPACITY:
                self.x["redone"] += 1
                cap = max(cap, self.x["cap"])
                while cap < need.value: cap *= 2
                self.x["cap"] = cap
                gathered = self._gather(hits, cap)
                continue
            assert st == 0
            self.x["keep"] = (out, starts, offs)
            C.memmove(C.addressof(hits_p._obj), C.addressof(C.c_void_p(out.ctypes.data)), C.sizeof(C.c_void_p))
            n_p._obj.value = int(starts[w])
            C.memmove(C.addressof(starts_p._obj), C.addressof(C.c_void_p(C.addressof(starts))), C.sizeof(C.c_void_p))
            C.memmove(C.addressof(offs_p._obj), C.addressof(C.c_void_p(C.addressof(offs))), C.sizeof(C.c_void_p))
            return 0
        return _lib.ERR_CAPACITY

rng = np.random.default_rng(100 + rank)
def mk(n, nrec):
    h = np.zeros(n, dtype=dist.HIT_DTYPE)
    h["pos"] = rng.integers(0, 1 << 30, n); h["record"] = rng.integers(0, nrec, n); h["pattern"] = rng.integers(0, 8, n)
    return h
nrec = 3 if rank == 0 else 2
mine = mk(40 if rank == 0 else 9, nrec)          # uneven: rank 0 holds more than four times rank 1's
ref_all, ref_off = dist.allgather_hits(mine, nrec)                          # the plain two-collective form: the reference
x = dist.HitExchanger(cap_hits=64, native_lib=FakeExchangeLib())
assert x.native
sc = FakeScratch(mine); SCRATCHES[sc._h] = sc
x.set_record_counts([3, 2])
w0 = x.start_scratch(sc, nrec); x.finish(w0)
hits, ranges, offs = x.gathered()
assert np.array_equal(hits, ref_all) and offs == ref_off == [0, 3] and ranges == [(0, 40), (40, 49)] and x.redone == 0
# two in flight, ended in order; then a capacity only rank 0 exceeds: both ranks redo together inside finish()
sc2 = FakeScratch(mine[::-1].copy()); SCRATCHES[sc2._h] = sc2
wa, wb = x.start_scratch(sc, nrec), x.start_scratch(sc2, nrec)
x.finish(wa); a_hits = x.gathered()[0]
x.finish(wb); b_hits = x.gathered()[0]
assert np.array_equal(a_hits, ref_all) and not np.array_equal(b_hits, ref_all) and len(b_hits) == len(ref_all)
small = dist.HitExchanger(cap_hits=16, native_lib=FakeExchangeLib())
small.set_record_counts([3, 2])
small.finish(small.start_scratch(sc, nrec))
h2, r2, o2 = small.gathered()
assert small.redone == 1 and small.cap >= 40 and np.array_equal(h2, ref_all) and r2 == ranges and o2 == [0, 3]
small.finish(small.start_scratch(sc, nrec))
assert small.redone == 1 and np.array_equal(small.gathered()[0], ref_all)
# the synchronous form sizes the native exchange too (reserve), and verify_native compares the two forms on every rank
x3 = dist.HitExchanger(cap_hits=4, native_lib=FakeExchangeLib())
x3.scratch_hits = None
orig = dist.hits_from_scratch
dist.hits_from_scratch = lambda s: s.hits
try:
    assert x3.verify_native(sc, nrec) is True and x3.native and x3.native_verified
    assert x3._L.ipcr_exchange_capacity(None) >= 40
    # a native exchange that loses a record on ONE rank: every rank falls back together
    class Lossy(FakeExchangeLib):
        def ipcr_exchange_end(self, x_, t, hits_p, n_p, starts_p, offs_p):
            st = super().ipcr_exchange_end(x_, t, hits_p, n_p, starts_p, offs_p)
            if rank == 1: self.x["keep"][0][0]["pos"] ^= 1
            return st
    x4 = dist.HitExchanger(cap_hits=64, native_lib=Lossy())
    assert x4.verify_native(sc, nrec) is False and not x4.native and x4.native_verified is False
    x4.finish(x4.start_scratch(sc, nrec))                                   # the torch form serves from here on
    assert np.array_equal(x4.gathered()[0], ref_all)
finally:
    dist.hits_from_scratch = orig
tdist.barrier()
print("NATIVE_OK", rank)
'''


def test_two_rank_gloo_native_exchange_caller(tmp_path):
    """HitExchanger's NATIVE form (what a multi-GPU job runs by default: ipcr_exchange_begin / _end) under a world-size-2
    job: the library table is replaced by one whose collective is gloo and whose unpack is the real ipcr_exchange_unpack --
    uneven counts, two exchanges in flight, the lock-step redo when only one rank overflows, the capacity agreed through
    the synchronous form, and verify_native (all ranks fall back together when one rank's native result is wrong)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "native_worker.py"
    script.write_text(NATIVE_WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"NATIVE_OK {rank}" in out, out
