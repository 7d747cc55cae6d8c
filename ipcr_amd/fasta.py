"""Mirror of core/fasta for the scan path: record / rolling-chunk stream (host side lives in
ipcr_amd/csrc/fasta.cpp)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterator

from . import _lib


@dataclass
class Record:
    """fasta.Record -- core/fasta/reader.go"""
    ID: str
    Seq: bytes


def StreamChunks(path: str, chunkSize: int = 0, overlap: int = 0) -> Iterator[Record]:
    """fasta.StreamChunksPathCtx -- core/fasta/path_ctx.go:19-38: whole records when chunking is
    off, else rolling windows with IDs 'id:start-end'."""
    h = C.c_void_p()
    _lib.check(_lib.lib().ipcr_fasta_open(path.encode(), chunkSize, overlap, C.byref(h)))
    try:
        rid, seq, n, got = C.c_char_p(), C.c_void_p(), C.c_uint64(), C.c_int32()
        while True:
            _lib.check(_lib.lib().ipcr_fasta_next(h, C.byref(rid), C.byref(seq), C.byref(n), C.byref(got)))
            if not got.value:
                break
            yield Record(rid.value.decode(), C.string_at(seq.value, n.value) if n.value else b"")
    finally:
        _lib.lib().ipcr_fasta_close(h)
