"""Multi-GPU form of the path: one process per GPU, records sharded over ranks, and one
all-gatherv of verified hit records (RCCL over xGMI on GPUs; gloo on CPU for tests).

The reference has no distributed mode; its unit of parallelism is the independent FASTA
record/chunk (internal/pipeline/pipeline.go:60-125).  Here every rank scans its own records
with the whole panel -- no data-path collective -- and only the hit records (32 B each, tens of
KB per genome) are exchanged, after which the amplicon join (core/engine/engine.go:108-404)
runs on the gathered list.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

# layout of ipcr_hit (include/ipcr_hip.h)
HIT_DTYPE = np.dtype([("pos", "<u8"), ("record", "<u4"), ("pattern", "<u4"), ("mm0", "<u8"), ("mm1", "<u8")])
assert HIT_DTYPE.itemsize == 32


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: Optional[str] = None):
    """Join the job.  backend 'nccl' IS RCCL on ROCm; 'gloo' for CPU tests."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    # IPCR_EXCHANGE_SELFTEST=1: a one-rank job still joins a group and runs the exchange (RCCL on one GPU)
    if (world > 1 or os.environ.get("IPCR_EXCHANGE_SELFTEST")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # no device_id: with it every all-gather of the per-step exchange costs 15 % of a step (measured with the
        # one-rank RCCL self-test), without it 2.5 %
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local, backend


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced split of records (or genomes) over ranks."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def hits_from_scratch(scratch) -> np.ndarray:
    """Copy of the ipcr_hit array of the last scan on `scratch` as a HIT_DTYPE array."""
    import ctypes as C
    ptr, n = scratch.raw_hits()
    if n == 0:
        return np.zeros(0, dtype=HIT_DTYPE)
    buf = (C.c_uint8 * (n * 32)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=HIT_DTYPE, count=n).copy()


def pattern_shard(cp, rank: Optional[int] = None, world: Optional[int] = None):
    """Pattern-axis sharding (one genome x a huge panel): this rank's panel object scans every world-th
    distinct pattern; all ranks scan the SAME records.  Gather the hits with `allgather_hits(..., same_records=True)`
    (or HitExchanger(same_records=True)) and join them with a full panel: the orientations of a pair are
    independent until the per-pair join (core/engine/compiled.go:192-207,260-265)."""
    r, w, _ = env_rank()
    rank = r if rank is None else rank
    world = w if world is None else world
    if world > 1:
        cp.set_shard(rank, world)
    return cp


def allgather_hits(local: np.ndarray, n_local_records: int, device=None, group=None, same_records: bool = False):
    """All-gatherv of hit records.  Returns (hits, record_offset_per_rank): every rank's hits
    concatenated in rank order with `record` rebased to a job-global record index (genome-parallel
    jobs), or left as it is when every rank scanned the same records (same_records: pattern shards)."""
    import torch
    import torch.distributed as dist
    assert local.dtype == HIT_DTYPE
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local.copy(), [0]
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    meta = torch.tensor([len(local), n_local_records], dtype=torch.int64, device=dev)
    metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)          # collective 1: counts
    metas = metas.cpu().view(world, 2)
    counts = [int(c) for c in metas[:, 0]]
    nrecs = [int(c) for c in metas[:, 1]]
    cmax = max(max(counts), 1)
    send = torch.zeros(cmax * 32, dtype=torch.uint8, device=dev)   # max-padded payload
    if len(local):
        send[:len(local) * 32] = torch.from_numpy(local.view(np.uint8).reshape(-1)).to(dev)
    recv = torch.empty(world * cmax * 32, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)           # collective 2: records
    recv = recv.cpu().numpy().reshape(world, cmax * 32)
    offsets, parts, off = [], [], 0
    for r in range(world):
        offsets.append(off)
        part = recv[r, :counts[r] * 32].copy().view(HIT_DTYPE)
        part["record"] += np.uint32(off)
        parts.append(part)
        if not same_records:
            off += nrecs[r]
    return (np.concatenate(parts) if parts else np.zeros(0, dtype=HIT_DTYPE)), offsets


def allgather_record_meta(lens: Sequence[int], flags: Sequence[int], device=None, group=None):
    """Gather every rank's record lengths and flags (bit0 = holds a non-ACGTacgt byte, bit1 =
    scanned in the unprotected-rc mode) in job-global record order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(lens), list(flags)
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(lens)], dtype=torch.int64, device=dev)
    ns = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ns, n, group=group)
    ns = [int(v) for v in ns.cpu()]
    nmax = max(max(ns), 1)
    send = torch.zeros(nmax * 2, dtype=torch.int64, device=dev)
    if len(lens):
        send[:len(lens)] = torch.tensor(list(lens), dtype=torch.int64)
        send[nmax:nmax + len(flags)] = torch.tensor(list(flags), dtype=torch.int64)
    recv = torch.empty(world * nmax * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.cpu().view(world, 2, nmax)
    all_lens, all_flags = [], []
    for r in range(world):
        all_lens += [int(v) for v in recv[r, 0, :ns[r]]]
        all_flags += [int(v) for v in recv[r, 1, :ns[r]]]
    return all_lens, all_flags


class HitExchanger:
    """Persistent buffers for the per-step all-gatherv of hit records: ONE collective per step
    (a max-padded all-gather whose header carries this rank's hit and record counts).

    Lock step: no rank ever decides alone.  A rank whose hits exceed the current capacity still enters
    the collective (header with the true count + the first `cap` records); after the gather EVERY rank
    reads the same headers, sees the same overflow, and all of them regrow and redo that exchange
    together (`finish` for the overlapped forms, the loop in `allgather` for the synchronous one).
    A rank whose scratch's device hit buffer is smaller than the exchange capacity sends through a
    device staging buffer of the agreed size instead of the zero-copy view -- the collective's shape
    never depends on a local condition."""

    def __init__(self, device=None, cap_hits: int = 4096, group=None, same_records: bool = False, native_lib=None):
        """native_lib: an object with the ipcr_exchange_* entry points to use instead of libipcr_hip.so's (tests: the native
        caller of this class under a world-size-2 gloo job, tests/test_dist_gloo.py); with it the native form is used on
        any device."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.same_records = same_records     # pattern shards: every rank scanned the same records, no index rebase
        self.active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or bool(os.environ.get("IPCR_EXCHANGE_SELFTEST")))
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.device = device if device is not None else torch.device("cpu")
        self.cap = 0
        self.redone = 0                      # exchanges repeated because some rank overflowed the capacity
        self.device_path = self.device.type == "cuda"
        self._rec_counts = [0] * self.world
        self._alloc(cap_hits)
        self._native = None                  # ipcr_exchange handle (csrc/exchange.cpp): RCCL inside the library
        self._native_redone0 = 0
        self.native_verified = None          # verify_native: True / False once it has run
        self._L = native_lib
        if self.active and (native_lib is not None or (self.device.type == "cuda" and not os.environ.get("IPCR_EXCHANGE_TORCH"))):
            self._native_create(cap_hits)

    def _native_create(self, cap_hits: int) -> None:
        """The library's own all-gather (ncclAllGather straight out of the scratch's device hit buffer, the same one a
        Go or C++ host calls).  Collective: every rank constructs its exchanger at the same point.  Three agreements, each
        a MIN over the ranks, so that all of them use the native form or none does and NOBODY enters ncclCommInitRank
        alone: (1) ipcr_exchange_available -- everything ipcr_exchange_create can fail on locally (librccl, the device),
        checked before any rank creates anything; (2) rank 0's RCCL id, broadcast with an ok flag; (3) the result of
        ipcr_exchange_create itself."""
        import ctypes as C
        from . import _lib
        torch, dist = self.torch, self.dist
        L = self._lib()
        dev_index = self.device.index if self.device.index is not None else (torch.cuda.current_device() if self.device.type == "cuda" else 0)

        def all_min(v: int) -> int:
            t = torch.tensor([v], dtype=torch.int32, device=self.device)
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return int(t.item())

        if all_min(int(L.ipcr_exchange_available(int(dev_index)))) != 1:
            return
        buf = torch.zeros(1 + 128, dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            raw = C.create_string_buffer(128)
            if L.ipcr_exchange_unique_id(raw) == _lib.OK:
                buf[0] = 1
                buf[1:] = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).to(self.device)
        if self.world > 1:
            dist.broadcast(buf, src=0, group=self.group)
        got = bytes(buf.cpu().numpy().tobytes())
        if got[0] != 1:
            return
        handle = C.c_void_p()
        st = L.ipcr_exchange_create(got[1:], self.world, self.rank, int(dev_index), int(cap_hits), int(self.same_records), C.byref(handle))
        ok = int(st == _lib.OK)
        if all_min(ok) == 1:
            self._native = handle
        elif ok:
            L.ipcr_exchange_destroy(handle)

    def verify_native(self, scratch, n_local_records: int) -> bool:
        """Collective, once per job (bench.py: after the first synchronous exchange): the native exchange of `scratch`'s
        hits against the torch.distributed form of the same hits, on every rank -- the native path's multi-rank logic
        (per-rank counts from the gathered headers, record rebasing) has unit tests on the CPU and a fake-transport test
        on one GPU, but a pool of one-GPU boxes cannot run it over a real communicator of several ranks; the first job
        that can checks it against the path that the world-size-2 tests cover, and all ranks fall back together
        (MIN) if the two disagree anywhere."""
        if self._native is None:
            return False
        want, want_ranges, want_offs = self.allgather(hits_from_scratch(scratch), n_local_records)
        ok = 1
        try:
            self.finish(self.start_scratch(scratch, n_local_records))
            got, got_ranges, got_offs = self.gathered()
            if list(got_offs) != list(want_offs) or len(got_ranges) != len(want_ranges):
                ok = 0
            else:
                for (a0, a1), (b0, b1) in zip(got_ranges, want_ranges):
                    # (the device list is in append order and may hold a window twice -- the seed index reports it once per key --
                    # the host list is sorted and de-duplicated: compare the sets of records)
                    if set(map(bytes, got[a0:a1])) != set(map(bytes, want[b0:b1])):
                        ok = 0
        except Exception:  # noqa: BLE001
            ok = 0
        t = self.torch.tensor([ok], dtype=self.torch.int32, device=self.device)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        self.native_verified = bool(int(t.item()))
        if not self.native_verified:
            self.close()
        return self.native_verified

    def _lib(self):
        if self._L is None:
            from . import _lib
            self._L = _lib.lib()
        return self._L

    def _check(self, status: int) -> None:
        if status != 0:
            from . import _lib
            raise _lib.IpcrError(status, self._lib().ipcr_last_error().decode(errors="replace"))

    def close(self) -> None:
        if self._native is not None:
            self._lib().ipcr_exchange_destroy(self._native)
            self._native = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def native(self) -> bool:
        return self._native is not None

    def _native_counts(self) -> None:
        import ctypes as C
        arr = (C.c_uint32 * self.world)(*[int(c) for c in self._rec_counts])
        self._check(self._lib().ipcr_exchange_set_record_counts(self._native, arr))

    def _alloc(self, cap: int) -> None:
        torch = self.torch
        self.cap = cap
        n = (cap + 1) * 32
        pin = self.device.type == "cuda"
        self.h_send = torch.zeros(n, dtype=torch.uint8, pin_memory=pin)
        self.h_recv = torch.zeros(self.world * n, dtype=torch.uint8, pin_memory=pin)
        self._views, self._host_work, self._last = {}, None, None
        if self.device.type == "cuda":
            nb = 64 + cap * 32
            self.d_send = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self.d_recv = torch.zeros(self.world * n, dtype=torch.uint8, device=self.device)
            self.d_stage = torch.zeros(nb, dtype=torch.uint8, device=self.device)
            # two receive slots (+ the pinned copy of their headers): up to two exchanges may be in flight
            self.d_recv_dev = [torch.zeros(self.world * nb, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.h_hdr = [torch.zeros(self.world * 64, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._slot = 0
            self.h_recv_dev = torch.zeros(self.world * nb, dtype=torch.uint8, pin_memory=True)
        else:
            self.d_send, self.d_recv = self.h_send, self.h_recv

    # -- overlapped form: the collective runs while the host joins this rank's own records --------
    def start(self, local: np.ndarray, n_local_records: int):
        """Enqueue the all-gatherv of this step's hit records (H2D of the local records + one
        all-gather) and return at once.  More records than the capacity: the first `cap` go out with
        the true count in the header, and `finish` redoes the exchange on every rank."""
        assert local.dtype == HIT_DTYPE
        if not self.active:
            return None
        if self._host_work is not None:               # one send buffer: the previous host-path exchange must be done
            self._host_work.wait()
        n = min(len(local), self.cap)
        hs = self.h_send.numpy()
        hs[:16].view(np.int64)[:] = (len(local), n_local_records)
        if n:
            hs[32:32 + n * 32] = local[:n].view(np.uint8).reshape(-1)
        if self.d_send is not self.h_send:
            self.d_send.copy_(self.h_send, non_blocking=True)
        self._host_work = self.dist.all_gather_into_tensor(self.d_recv, self.d_send, group=self.group, async_op=True)
        return ("host", self._host_work, local, n_local_records, self.cap)

    # -- device form (RCCL): the hit records never visit the host on the sending side ---------------
    class _DevView:
        """zero-copy torch view of a raw device range (__cuda_array_interface__)"""
        def __init__(self, ptr: int, nbytes: int):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    def _view(self, ptr: int, nbytes: int):
        key = (ptr, nbytes)
        view = self._views.get(key)
        if view is None:
            view = self.torch.as_tensor(self._DevView(ptr, nbytes), device=self.device)
            self._views = {k: v for k, v in self._views.items() if k[0] != ptr}
            self._views[key] = view
        return view

    def agree_on_device_path(self, scratch) -> bool:
        """Every rank tries the zero-copy view of its scratch's device hit buffer (the one step of the device
        form that is not a plain collective); the ranks then take the minimum of their verdicts, so that all of them
        use the device form or all of them the host copy -- never a mix that would leave a collective half entered."""
        ok = 0
        if self.active and self.device.type == "cuda" and not os.environ.get("IPCR_EXCHANGE_HOST"):
            try:
                ptr, _, cap = scratch.device_hits()
                nbytes = 64 + min(cap, self.cap) * 32
                view = self.torch.as_tensor(self._DevView(ptr, nbytes), device=self.device)
                ok = int(view.numel() == nbytes and view.data_ptr() == ptr)
            except Exception:
                ok = 0
        if self.active:
            t = self.torch.tensor([ok], dtype=self.torch.int32, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
            ok = int(t.item())
        self.device_path = bool(ok)
        return self.device_path

    def start_scratch(self, scratch, n_local_records: int):
        """All-gather straight out of the scratch's device hit buffer (64-byte counter header + the first
        `cap` hit slots, ipcr_scratch_device_hits): no copy of the records to the host and back.  The
        buffer must stay untouched until finish(): do not begin the scratch's next scan before."""
        if not self.active:
            return None
        if self._native is not None:
            import ctypes as C
            ticket = C.c_int32(-1)
            self._check(self._lib().ipcr_exchange_begin(self._native, scratch._h, C.byref(ticket)))
            return ("native", ticket.value, scratch, n_local_records)
        if self.device.type != "cuda" or not self.device_path:
            return self.start(hits_from_scratch(scratch), n_local_records)
        nbytes = 64 + self.cap * 32
        from . import _lib
        try:
            ptr, n, cap = scratch.device_hits()
        except _lib.IpcrError:
            # a capped scan that ran in segments keeps its hits on the host (ipcr_scratch_device_hits refuses: the device
            # buffer holds the last range only): the same block shape -- header with the true count, the first `cap`
            # records -- rebuilt in the staging buffer, so the collective this rank enters is the one the others enter
            local = hits_from_scratch(scratch)
            blk = np.zeros(nbytes, dtype=np.uint8)
            blk[:64].view(np.uint64)[1] = len(local)
            m = min(len(local), self.cap)
            if m:
                blk[64:64 + m * 32] = local[:m].view(np.uint8).reshape(-1)
            self.d_stage.copy_(self.torch.from_numpy(blk))
            ptr, cap, send = None, 0, self.d_stage
        if ptr is None:
            pass
        elif cap >= self.cap:
            send = self._view(ptr, nbytes)
        else:   # this rank's buffer is smaller than the agreed shape: same bytes through a staging buffer
            m = 64 + cap * 32
            self.d_stage[:m].copy_(self._view(ptr, m), non_blocking=True)
            send = self.d_stage
        slot = self._slot = (self._slot + 1) & 1
        recv, hdr = self.d_recv_dev[slot], self.h_hdr[slot]
        work = self.dist.all_gather_into_tensor(recv[: self.world * nbytes], send, group=self.group, async_op=True)
        work.wait()                                   # current stream waits (no host block) ...
        hdr.view(self.world, 64).copy_(recv.view(self.world, nbytes)[:, :64], non_blocking=True)  # every rank's counters
        ev = self.torch.cuda.Event()
        ev.record()                                   # ... so this event completes exactly when gather + header copy have
        return ("dev", work, ev, recv, hdr, self.cap, scratch, n_local_records)

    @staticmethod
    def _device_counts(hdr_bytes: np.ndarray, world: int) -> np.ndarray:
        hdr = hdr_bytes.reshape(world, 64).view(np.uint64).reshape(world, 8)
        return np.maximum(hdr[:, 1], hdr[:, 5]).astype(np.int64)   # the counter set of the last scan is the non-zero one

    def finish(self, work) -> None:
        """Wait until every rank's hit records of this step are resident in this rank's memory.  If the
        gathered headers show that some rank had more hits than the capacity, every rank (all read the
        same headers) regrows and repeats that exchange synchronously; `gathered()` then returns the
        complete result."""
        if work is None:
            return
        if work[0] == "native":
            import ctypes as C
            from . import _lib
            L = self._lib()
            hits, n = C.POINTER(_lib.Hit)(), C.c_int64(0)
            starts, offs = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
            self._check(L.ipcr_exchange_end(self._native, work[1], C.byref(hits), C.byref(n), C.byref(starts), C.byref(offs)))
            if n.value:
                raw = (C.c_uint8 * (n.value * 32)).from_address(C.addressof(hits.contents))
                arr = np.frombuffer(raw, dtype=HIT_DTYPE, count=n.value).copy()
            else:
                arr = np.zeros(0, dtype=HIT_DTYPE)
            ranges = [(int(starts[r]), int(starts[r + 1])) for r in range(self.world)]
            self._last = ("full", arr, ranges, [int(offs[r]) for r in range(self.world)])
            self.redone = self._native_redone0 + int(L.ipcr_exchange_redone(self._native))
            self.cap = max(self.cap, int(L.ipcr_exchange_capacity(self._native)))
            return
        if work[0] == "dev":
            _, _, ev, recv, hdr, cap, scratch, nrec = work
            ev.synchronize()
            counts = self._device_counts(hdr.numpy(), self.world)
            if int(counts.max()) > cap:
                self.redone += 1
                # size_hint: the RAW device count (duplicates included) every rank has just read -- the de-duplicated host
                # list may fit a capacity the device buffer does not, and the next device-form exchange would overflow again
                self._last = ("full",) + tuple(self.allgather(hits_from_scratch(scratch), nrec, size_hint=int(counts.max())))
            else:
                self._last = ("dev", recv, cap, counts)
            return
        _, w, local, nrec, cap = work
        w.wait()
        if self._host_work is w:
            self._host_work = None
        if self.device.type == "cuda":
            self.h_recv.copy_(self.d_recv, non_blocking=True)
            self.torch.cuda.current_stream(self.device).synchronize()
        hr = self.h_recv.numpy().reshape(self.world, (cap + 1) * 32)
        meta = hr[:, :16].copy().view(np.int64).reshape(self.world, 2)
        if int(meta[:, 0].max()) > cap:
            self.redone += 1
            self._last = ("full",) + tuple(self.allgather(local, nrec))
        else:
            self._last = ("full",) + tuple(self._unpack(hr, meta))

    def gathered(self):
        """Result of the exchange most recently finished -> same triple as allgather()."""
        if not self.active:
            raise RuntimeError("no exchange in a single-process job")
        if self._last is None:
            raise RuntimeError("no finished exchange")
        if self._last[0] == "full":
            return self._last[1:]
        _, recv, cap, counts = self._last
        nbytes = 64 + cap * 32
        host = self.h_recv_dev if self.h_recv_dev.numel() == recv.numel() else self.torch.empty(recv.numel(), dtype=self.torch.uint8, pin_memory=True)
        host.copy_(recv)
        return self._unpack_device(host.numpy().reshape(self.world, nbytes), cap, counts)

    def set_record_counts(self, counts) -> None:
        """records per rank (static for a job; the device form does not resend them every step)"""
        self._rec_counts = [int(c) for c in counts]
        if self._native is not None:
            self._native_counts()

    def _unpack_device(self, hr: np.ndarray, cap: int, counts: np.ndarray):
        parts, ranges, offsets, off, pos = [], [], [], 0, 0
        for r in range(self.world):
            cnt = int(counts[r])
            assert cnt <= cap, f"rank {r} sent {cnt} hits through an exchange of capacity {cap}"
            part = hr[r, 64:64 + cnt * 32].copy().view(HIT_DTYPE)
            part["record"] += np.uint32(off)
            parts.append(part)
            ranges.append((pos, pos + cnt))
            offsets.append(off)
            pos += cnt
            if not self.same_records:
                off += self._rec_counts[r]
        return np.concatenate(parts), ranges, offsets

    def _unpack(self, hr: np.ndarray, meta: np.ndarray):
        parts, ranges, offsets, off, pos = [], [], [], 0, 0
        for r in range(self.world):
            cnt, nrec = int(meta[r, 0]), int(meta[r, 1])
            part = hr[r, 32:32 + cnt * 32].copy().view(HIT_DTYPE)
            part["record"] += np.uint32(off)
            parts.append(part)
            ranges.append((pos, pos + cnt))
            offsets.append(off)
            pos += cnt
            if not self.same_records:
                off += nrec
        return np.concatenate(parts), ranges, offsets

    def allgather(self, local: np.ndarray, n_local_records: int, size_hint: int = 0):
        """-> (all hits with job-global record index, per-rank (hit_start, hit_end), record offsets).
        size_hint: a count the capacity should also cover on every rank, e.g. the number of records in the
        scratch's DEVICE hit buffer (the seed-index filter leaves duplicates there that `local` no longer has):
        the overlapped device form then fits without a redo."""
        assert local.dtype == HIT_DTYPE
        if not self.active:
            return local, [(0, len(local))], [0]
        torch = self.torch
        if self._host_work is not None:
            self._host_work.wait()
            self._host_work = None
        while True:
            n = min(len(local), self.cap)
            hs = self.h_send.numpy()
            hs[:24].view(np.int64)[:] = (len(local), n_local_records, max(int(size_hint), len(local)))
            if n:
                hs[32:32 + n * 32] = local[:n].view(np.uint8).reshape(-1)
            if self.d_send is not self.h_send:
                self.d_send.copy_(self.h_send, non_blocking=True)
            self.dist.all_gather_into_tensor(self.d_recv, self.d_send, group=self.group)
            if self.d_recv is not self.h_recv:
                self.h_recv.copy_(self.d_recv, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()
            hr = self.h_recv.numpy().reshape(self.world, (self.cap + 1) * 32)
            meta3 = hr[:, :24].copy().view(np.int64).reshape(self.world, 3)
            meta = np.ascontiguousarray(meta3[:, :2])
            need = int(max(meta3[:, 0].max(), meta3[:, 2].max()))
            self._rec_counts = [int(c) for c in meta[:, 1]]
            if self._native is not None:   # every rank has read the same `need`: the native exchange is sized for it too
                self._native_counts()
                self._check(self._lib().ipcr_exchange_reserve(self._native, max(int(need), self.cap)))
            if need <= self.cap:
                break
            cap = self.cap
            while cap < need:
                cap *= 2
            rc = self._rec_counts
            self._alloc(cap)  # identical on every rank: all saw the same header
            self._rec_counts = rc
            self.redone_sync = getattr(self, "redone_sync", 0) + 1
        out = self._unpack(hr, meta)
        self._last = ("full",) + tuple(out)
        return out
