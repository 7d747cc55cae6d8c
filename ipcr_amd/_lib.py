"""ctypes binding of libipcr_hip.so (C ABI: include/ipcr_hip.h).

The library is the product; there is no Python or CPU fallback.  If it is missing the import
fails loudly and tells the user how to build it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("IPCR_HIP_LIBRARY") or os.path.join(_HERE, "libipcr_hip.so")  # override: A/B builds

IPCR_MAX_MM = 16
IPCR_MAX_PRIMER_LEN = 128

OK, ERR_INVALID, ERR_PRIMER, ERR_UNSUPPORTED, ERR_DEVICE, ERR_CAPACITY, ERR_ABORTED = range(7)


class IpcrError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"ipcr_hip status {status}: {message}")
        self.status = status
        self.message = message


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_mm", "terminal_window", "min_len", "max_len", "hit_cap",
                                         "seed_len", "circular", "need_sites")]


class Pair(C.Structure):
    _fields_ = [("id", C.c_char_p), ("forward", C.c_char_p), ("reverse", C.c_char_p),
                ("min_product", C.c_int32), ("max_product", C.c_int32)]


class Product(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("length", C.c_int64), ("pair", C.c_int32),
                ("record", C.c_int32), ("type", C.c_int32), ("fwd_mm", C.c_int32), ("rev_mm", C.c_int32),
                ("n_fwd_idx", C.c_int32), ("n_rev_idx", C.c_int32),
                ("fwd_idx", C.c_uint8 * IPCR_MAX_MM), ("rev_idx", C.c_uint8 * IPCR_MAX_MM)]


class Hit(C.Structure):
    _fields_ = [("pos", C.c_uint64), ("record", C.c_uint32), ("pattern", C.c_uint32),
                ("mm_mask", C.c_uint64 * 2)]


class ProbeHit(C.Structure):
    _fields_ = [("found", C.c_int32), ("strand", C.c_int32), ("pos", C.c_int32), ("mm", C.c_int32)]


class ChunkWindow(C.Structure):
    """ipcr_chunk_window: one rolling window of a record (ipcr_scan_genome_chunked)"""
    _fields_ = [("record", C.c_uint32), ("plain", C.c_uint32), ("start", C.c_uint64), ("end", C.c_uint64),
                ("reset", C.c_uint32), ("reserved0", C.c_uint32)]


class Window(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("record", C.c_int32), ("reserved", C.c_int32)]


class NestedHit(C.Structure):
    _fields_ = [("found", C.c_int32), ("pair", C.c_int32), ("type", C.c_int32), ("fwd_mm", C.c_int32),
                ("rev_mm", C.c_int32), ("reserved", C.c_int32), ("start", C.c_int64), ("end", C.c_int64),
                ("length", C.c_int64)]


class ScanStats(C.Structure):
    _fields_ = [("pack_ms", C.c_double), ("filter_ms", C.c_double), ("verify_ms", C.c_double),
                ("total_ms", C.c_double), ("bases", C.c_uint64), ("tile_bytes", C.c_uint64),
                ("candidates", C.c_uint64), ("hits", C.c_uint64), ("products", C.c_uint64),
                ("kernel_kind", C.c_int32), ("n_patterns", C.c_int32), ("enqueue_ms", C.c_double),
                ("wait_ms", C.c_double), ("sort_ms", C.c_double), ("join_ms", C.c_double),
                ("handover_refetched", C.c_uint64), ("handover_checked", C.c_uint64),
                ("handover_check_diffs", C.c_uint64), ("leftover_patterns", C.c_uint32), ("leftover_kernels", C.c_uint32),
                ("hostpack_ms", C.c_double), ("segmented", C.c_uint32), ("pattern_set", C.c_uint32)]


EMIT_FN = C.CFUNCTYPE(C.c_int, C.POINTER(Product), C.c_void_p)

# every symbol include/ipcr_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ipcr_version": (C.c_char_p, []),
    "ipcr_last_error": (C.c_char_p, []),
    "ipcr_set_device": (C.c_int, [C.c_int]),
    "ipcr_device_count": (C.c_int, []),
    "ipcr_bind_thread_to_device": (C.c_int, [C.c_int]),
    "ipcr_iupac_mask": (C.c_uint8, [C.c_uint8]),
    "ipcr_base_match": (C.c_int, [C.c_uint8, C.c_uint8]),
    "ipcr_revcomp": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p]),
    "ipcr_panel_create": (C.c_int, [C.POINTER(Config), C.POINTER(Pair), C.c_int32, C.POINTER(C.c_void_p)]),
    "ipcr_panel_destroy": (None, [C.c_void_p]),
    "ipcr_panel_num_pairs": (C.c_int32, [C.c_void_p]),
    "ipcr_panel_num_patterns": (C.c_int32, [C.c_void_p]),
    "ipcr_panel_max_primer_len": (C.c_int32, [C.c_void_p]),
    "ipcr_panel_have": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char]),
    "ipcr_panel_set_specialize": (C.c_int, [C.c_void_p, C.c_int32]),
    "ipcr_panel_set_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "ipcr_panel_scanned_patterns": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "ipcr_panel_filter_source": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ipcr_panel_num_patterns_total": (C.c_int32, [C.c_void_p]),
    "ipcr_panel_pattern_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ipcr_panel_slot_pattern": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char, C.c_int32]),
    "ipcr_scratch_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ipcr_scratch_create_on": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "ipcr_scratch_device": (C.c_int32, [C.c_void_p]),
    "ipcr_panel_device_slots": (C.c_int32, [C.c_void_p]),
    "ipcr_panel_wait_ready": (C.c_int, [C.c_void_p]),
    "ipcr_genome_create_on": (C.c_int, [C.c_uint64, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]),
    "ipcr_genome_device": (C.c_int32, [C.c_void_p]),
    "ipcr_scratch_create_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ipcr_scratch_destroy": (None, [C.c_void_p]),
    "ipcr_scratch_stats": (C.c_int, [C.c_void_p, C.POINTER(ScanStats)]),
    "ipcr_scratch_products": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(Product)), C.POINTER(C.c_int64)]),
    "ipcr_scratch_hits": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(Hit)), C.POINTER(C.c_int64)]),
    "ipcr_scratch_device_hits": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ipcr_scan_chunk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "ipcr_pack_ascii": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ipcr_genome_create": (C.c_int, [C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ipcr_genome_destroy": (None, [C.c_void_p]),
    "ipcr_genome_add_record": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64]),
    "ipcr_genome_add_record_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "ipcr_lcg_fill_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64]),
    "ipcr_genome_read": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint64]),
    "ipcr_genome_num_records": (C.c_uint32, [C.c_void_p]),
    "ipcr_genome_record_len": (C.c_uint64, [C.c_void_p, C.c_uint32]),
    "ipcr_genome_record_id": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "ipcr_genome_total_bases": (C.c_uint64, [C.c_void_p]),
    "ipcr_genome_tile_bytes": (C.c_uint64, [C.c_void_p]),
    "ipcr_genome_pack_ms": (C.c_double, [C.c_void_p]),
    "ipcr_genome_record_flags": (C.c_uint8, [C.c_void_p, C.c_uint32]),
    "ipcr_fasta_open": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "ipcr_fasta_close": (None, [C.c_void_p]),
    "ipcr_fasta_next": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "ipcr_genome_add_fasta": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ipcr_scan_genome": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipcr_scan_genome_hits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipcr_scan_genome_chunked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "ipcr_scratch_chunk_windows": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(ChunkWindow)), C.POINTER(C.c_int64)]),
    "ipcr_chunk_windows": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.POINTER(ChunkWindow), C.c_int64, C.POINTER(C.c_int64)]),
    "ipcr_scan_genome_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipcr_scratch_chain_after": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ipcr_scan_genome_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipcr_join_hits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint8), C.c_uint32, C.c_void_p, C.c_void_p]),
    "ipcr_exchange_available": (C.c_int32, [C.c_int32]),
    "ipcr_exchange_unpack": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_uint32), C.c_int32, C.c_void_p, C.c_uint64,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "ipcr_exchange_unique_id": (C.c_int, [C.c_char_p]),
    "ipcr_exchange_create": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]),
    "ipcr_exchange_destroy": (None, [C.c_void_p]),
    "ipcr_exchange_set_records": (C.c_int, [C.c_void_p, C.c_uint32]),
    "ipcr_exchange_set_record_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "ipcr_exchange_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "ipcr_exchange_end": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(Hit)), C.POINTER(C.c_int64),
                                    C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]),
    "ipcr_exchange_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "ipcr_exchange_capacity": (C.c_uint64, [C.c_void_p]),
    "ipcr_exchange_redone": (C.c_uint64, [C.c_void_p]),
    "ipcr_probe_best_hit": (C.c_int, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_int32, C.POINTER(ProbeHit)]),
    "ipcr_probe_products": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(ProbeHit), C.c_int64]),
    "ipcr_probe_products_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32]),
    "ipcr_probe_products_end": (C.c_int, [C.c_void_p, C.POINTER(ProbeHit), C.c_int64]),
    "ipcr_probe_scratch_products_begin": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "ipcr_probe_scratch_products": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(ProbeHit), C.c_int64]),
    "ipcr_nested_windows": (C.c_int, [C.c_void_p, C.POINTER(Window), C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(NestedHit)]),
    "ipcr_nested_products": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NestedHit), C.c_int64]),
}

_lib = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 /
    libhiprtc; if this library pulled in /opt/rocm's copies first, a later `import torch`
    would load a second runtime and find no GPU.  So when torch is installed, load ITS copies
    first (same SONAMEs, so libipcr_hip.so binds to them); otherwise /opt/rocm's are used."""
    import importlib.util
    if os.environ.get("IPCR_HIP_RUNTIME") == "system":   # a process that will never import torch: /opt/rocm's runtime
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libamdhip64.so", "libhiprtc.so"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C ipcr_amd/csrc). "
                "ipcr_amd has no CPU fallback.")
        _preload_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != OK:
        raise IpcrError(status, lib().ipcr_last_error().decode(errors="replace"))
