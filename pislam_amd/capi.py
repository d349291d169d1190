"""ctypes binding of libpislam_hip.so (include/pislam_hip.h).

Fails loudly: if the shared library is missing or cannot be loaded an
ImportError/OSError propagates — there is no Python or CPU fallback for any
entry point.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import build as _build

PISLAM_OK = 0
ABI_VERSION = 2

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


class Level(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("row0", ctypes.c_int32), ("col0", ctypes.c_int32)]


class FrontendParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "vstep", "rows", "nlevels", "border", "fast_threshold", "harris_threshold",
        "log_bucket_size", "bucket_limit", "words", "max_keypoints")]


# name -> (restype, argtypes); every symbol declared in include/pislam_hip.h
SYMBOLS = {
    "pislam_abi_version": (_i, []),
    "pislam_ctx_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "pislam_ctx_destroy": (_i, [_vp]),
    "pislam_ctx_set_stream": (_i, [_vp, _vp]),
    "pislam_ctx_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "pislam_ctx_synchronize": (_i, [_vp]),
    "pislam_last_error": (ctypes.c_char_p, [_vp]),
    "pislam_fast_detect": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i]),
    "pislam_fast_score_harris": (_i, [_vp, _i, _i, _i, _i, _vp, ctypes.c_int32, _vp]),
    "pislam_fast_extract": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, ctypes.POINTER(_sz)]),
    "pislam_orb_compute": (_i, [_vp, _i, _i, _vp, _vp, _sz, _vp]),
    "pislam_harris_score_points": (_i, [_vp, _i, _vp, _vp, _sz, ctypes.c_int32, _vp]),
    "pislam_centroids_size": (_sz, [_sz]),
    "pislam_orb_centroids": (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    "pislam_orb_angles": (_i, [_vp, _vp, _sz, _vp]),
    "pislam_brief_describe": (_i, [_vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pislam_brief_table": (ctypes.POINTER(ctypes.c_int8), []),
    "pislam_gaussian5x5": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "pislam_bilinear7_8": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "pislam_bilinear13_16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "pislam_pyramid_layout": (_i, [_i, _i, _i, ctypes.POINTER(ctypes.c_int32), _i, ctypes.POINTER(Level),
                                   ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "pislam_pyramid_build_batch": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(Level), _vp, _i, _sz,
                                        _i, _vp, _i, _i, _sz, _i]),
    "pislam_orb_frontend_batch": (_i, [_vp, ctypes.POINTER(FrontendParams), ctypes.POINTER(Level), _vp,
                                       _sz, _i, _vp, _vp, _vp]),
    "pislam_frontend_reserve": (_i, [_vp, ctypes.POINTER(FrontendParams), ctypes.POINTER(Level), _i]),
    "pislam_frontend_get_score_map": (_i, [_vp, _i, _vp]),
    "pislam_frontend_last_timing": (_i, [_vp, ctypes.POINTER(ctypes.c_float),
                                         ctypes.POINTER(ctypes.c_float * 3)]),
    "pislam_frontend_last_stats": (_i, [_vp, ctypes.POINTER(ctypes.c_uint32 * 2)]),
    "pislam_frontend_last_path": (ctypes.c_uint, [_vp]),
    "pislam_debug_build_plan": (_i, [ctypes.POINTER(FrontendParams), ctypes.POINTER(Level), _i, _i, _i, ctypes.c_char_p,
                                     ctypes.POINTER(ctypes.c_uint32 * 8), ctypes.c_char_p, _sz]),
    "pislam_pipeline_create": (_i, [_i, _i, ctypes.POINTER(_vp)]),
    "pislam_pipeline_destroy": (_i, [_vp]),
    "pislam_pipeline_depth": (_i, [_vp]),
    "pislam_pipeline_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "pislam_pipeline_reserve": (_i, [_vp, ctypes.POINTER(FrontendParams), ctypes.POINTER(Level), _i]),
    "pislam_pipeline_submit": (_i, [_vp, ctypes.POINTER(FrontendParams), ctypes.POINTER(Level), _vp, _sz, _i, _vp, _vp, _vp,
                                    _vp, _i, ctypes.POINTER(ctypes.c_uint64)]),
    "pislam_pipeline_wait": (_i, [_vp, ctypes.c_uint64, _vp]),
    "pislam_pipeline_synchronize": (_i, [_vp]),
    "pislam_pipeline_stats": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64 * 4)]),
    "pislam_pipeline_stream": (_vp, [_vp, ctypes.c_uint64]),
    "pislam_pipeline_lane": (_vp, [_vp, _i]),
    "pislam_pipeline_last_error": (ctypes.c_char_p, [_vp]),
    "pislam_match_hamming": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "pislam_match_hamming_batch": (_i, [_vp, _i, _vp, _vp, _sz, _vp, _vp, _sz, _i, _vp, _vp, _vp]),
    "pislam_dist_shard": (_i, [_i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "pislam_dist_get_unique_id": (_i, [ctypes.c_char_p]),
    "pislam_dist_init": (_i, [_vp, ctypes.c_char_p, _i, _i]),
    "pislam_dist_rank": (_i, [_vp]),
    "pislam_dist_world": (_i, [_vp]),
    "pislam_dist_comm_count": (_i, [_vp]),
    "pislam_dist_allgather_counts": (_i, [_vp, _vp, _sz, _vp]),
    "pislam_dist_fence": (_i, [_vp, _i]),
    "pislam_dist_allgather_counts_on": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "pislam_dist_fence_on": (_i, [_vp, _i, _vp]),
    "pislam_debug_shader_clock": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_double)]),
    "pislam_dist_synchronize": (_i, [_vp]),
    "pislam_dist_allreduce_max": (_i, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "pislam_dist_finalize": (_i, [_vp]),
}

_LIB = None


def library_path() -> str:
    return _build.LIB


def load(rebuild_if_stale: bool = True):
    """dlopen the in-tree library (building it first if hipcc and sources say it is stale)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: torch bundles its own libamdhip64 (soname libamdhip64.so.7) and
    # asks for it as "libamdhip64.so", so it must be loaded BEFORE this library (which then binds to
    # the already-loaded runtime and can use torch's device pointers in place).  Loading us first
    # would bring in /opt/rocm's copy as a second runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _build.check_override()              # PISLAM_HIP_LIB (development A/B): must exist, announced once
    path = _build.LIB
    if rebuild_if_stale:
        try:
            path = _build.build()
        except Exception:
            if not os.path.exists(path):
                raise
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m pislam_amd.build` (hipcc, gfx950). "
                          "pislam_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype, fn.argtypes = res, args
    if lib.pislam_abi_version() != ABI_VERSION:
        raise ImportError("libpislam_hip.so ABI version mismatch")
    _LIB = lib
    return lib


class PislamError(RuntimeError):
    pass


def dist_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI (call on one rank, hand the bytes to all)."""
    buf = ctypes.create_string_buffer(128)
    rc = load().pislam_dist_get_unique_id(buf)
    if rc != PISLAM_OK:
        raise PislamError(f"pislam_dist_get_unique_id failed ({rc}): RCCL not loadable?")
    return buf.raw


def dist_shard(global_batch: int, rank: int, world: int):
    first, count = _i(0), _i(0)
    rc = load().pislam_dist_shard(global_batch, rank, world, ctypes.byref(first), ctypes.byref(count))
    if rc != PISLAM_OK:
        raise PislamError(f"pislam_dist_shard failed ({rc})")
    return first.value, first.value + count.value


def ptr(a) -> int:
    """Raw address of a numpy array (host) or a torch tensor (host or device)."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        if not a.flags.c_contiguous:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return a.data_ptr()
    if isinstance(a, int):
        return a
    raise TypeError(f"cannot take the address of {type(a)}")


class Context:
    """Owns one pislam_ctx (one device, one stream)."""

    def __init__(self, device: int = -1, stream: int | None = None):
        self.lib = load()
        h = _vp()
        rc = self.lib.pislam_ctx_create(device, ctypes.byref(h))
        if rc != PISLAM_OK:
            raise PislamError(f"pislam_ctx_create failed ({rc}): no usable HIP device? "
                              "(there is no CPU fallback)")
        self.h = h
        if stream is not None:
            self.set_stream(stream)

    @classmethod
    def borrowed(cls, handle):
        """A Context object over a pislam_ctx someone else owns (a pipeline lane): never destroyed from here."""
        c = cls.__new__(cls)
        c.lib = load()
        c.h = _vp(handle)
        c.owned = False
        return c

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "owned", True):
                self.lib.pislam_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str):
        if rc != PISLAM_OK:
            msg = self.lib.pislam_last_error(self.h)
            raise PislamError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_stream(self, stream: int):
        self.check(self.lib.pislam_ctx_set_stream(self.h, _vp(stream)), "pislam_ctx_set_stream")

    # ---- multi-GPU (pislam_dist_*): one process per GPU, one count all-gather over RCCL ----
    def dist_init(self, unique_id: bytes | None, rank: int, world: int):
        self.check(self.lib.pislam_dist_init(self.h, unique_id, rank, world), "pislam_dist_init")

    def dist_allgather_counts(self, local_counts, all_counts, stream: int | None = None):
        """Device tensors: all_counts[r*n + i] = rank r's local_counts[i]; asynchronous (collective stream),
        ordered after the context stream — or after `stream` (a raw hipStream_t: another pipeline's)."""
        if stream is None:
            self.check(self.lib.pislam_dist_allgather_counts(self.h, ptr(local_counts), local_counts.numel(),
                                                             ptr(all_counts)), "pislam_dist_allgather_counts")
        else:
            self.check(self.lib.pislam_dist_allgather_counts_on(self.h, _vp(stream), ptr(local_counts),
                                                                local_counts.numel(), ptr(all_counts)),
                       "pislam_dist_allgather_counts_on")

    def dist_fence(self, back: int = 1, stream: int | None = None):
        """The context stream — or `stream` (a raw hipStream_t) — waits on the device for the collective issued
        `back` exchanges ago."""
        if stream is None:
            self.check(self.lib.pislam_dist_fence(self.h, back), "pislam_dist_fence")
        else:
            self.check(self.lib.pislam_dist_fence_on(self.h, back, _vp(stream)), "pislam_dist_fence_on")

    def shader_clock_ghz(self, micros: int = 200) -> float:
        v = ctypes.c_double(0)
        self.check(self.lib.pislam_debug_shader_clock(self.h, micros, ctypes.byref(v)), "pislam_debug_shader_clock")
        return float(v.value)

    def dist_synchronize(self):
        self.check(self.lib.pislam_dist_synchronize(self.h), "pislam_dist_synchronize")

    def dist_allreduce_max(self, value: float) -> float:
        v = ctypes.c_double(value)
        self.check(self.lib.pislam_dist_allreduce_max(self.h, ctypes.byref(v)), "pislam_dist_allreduce_max")
        return float(v.value)

    def dist_comm_count(self) -> int:
        """Ranks RCCL itself reports for this context's communicator (ncclCommCount); 0 without one."""
        n = self.lib.pislam_dist_comm_count(self.h)
        if n < 0:
            self.check(n, "pislam_dist_comm_count")
        return n

    def dist_finalize(self):
        self.check(self.lib.pislam_dist_finalize(self.h), "pislam_dist_finalize")

    def set_option(self, key: str, value: int):
        self.check(self.lib.pislam_ctx_set_option(self.h, key.encode(), int(value)), f"set_option({key})")

    def synchronize(self):
        self.check(self.lib.pislam_ctx_synchronize(self.h), "pislam_ctx_synchronize")


class Pipeline:
    """pislam_pipeline_*: `depth` contexts behind one object, batch k on lane k % depth (batches in flight)."""

    def __init__(self, device: int = -1, depth: int = 3):
        self.lib = load()
        h = _vp()
        rc = self.lib.pislam_pipeline_create(device, depth, ctypes.byref(h))
        if rc != PISLAM_OK:
            raise PislamError(f"pislam_pipeline_create failed ({rc})")
        self.h, self.depth = h, depth

    def check(self, rc: int, what: str):
        if rc != PISLAM_OK:
            msg = self.lib.pislam_pipeline_last_error(self.h)
            raise PislamError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_option(self, key: str, value: int):
        self.check(self.lib.pislam_pipeline_set_option(self.h, key.encode(), int(value)), f"pipeline set_option({key})")

    def reserve(self, params, levels, batch: int):
        self.check(self.lib.pislam_pipeline_reserve(self.h, ctypes.byref(params), levels, batch), "pislam_pipeline_reserve")

    def submit(self, params, levels, pyramids, kp, desc, counts, input_stream: int | None = None) -> int:
        t = ctypes.c_uint64(0)
        stride = int(pyramids.stride(0))
        self.check(self.lib.pislam_pipeline_submit(self.h, ctypes.byref(params), levels, ptr(pyramids), stride,
                                                   int(pyramids.shape[0]), ptr(kp), ptr(desc), ptr(counts),
                                                   _vp(input_stream or 0), 0 if input_stream is None else 1,
                                                   ctypes.byref(t)), "pislam_pipeline_submit")
        return int(t.value)

    def wait(self, ticket: int, stream: int):
        self.check(self.lib.pislam_pipeline_wait(self.h, ticket, _vp(stream)), "pislam_pipeline_wait")

    def stream_of(self, ticket: int) -> int:
        return int(self.lib.pislam_pipeline_stream(self.h, ticket) or 0)

    def lane(self, i: int) -> "Context":
        """The context of lane i (owned by the pipeline): for work that has to run on that lane's stream
        (a pyramid build in front of a batch, statistics)."""
        h = self.lib.pislam_pipeline_lane(self.h, i)
        if not h:
            raise PislamError("no such lane")
        return Context.borrowed(h)

    def stats(self) -> dict:
        st = (ctypes.c_uint64 * 4)()
        self.check(self.lib.pislam_pipeline_stats(self.h, ctypes.byref(st)), "pislam_pipeline_stats")
        return {"submitted": int(st[0]), "replayed_from_graphs": int(st[1]), "captured": int(st[2]), "capture_failed": int(st[3])}

    def synchronize(self):
        self.check(self.lib.pislam_pipeline_synchronize(self.h), "pislam_pipeline_synchronize")

    def close(self):
        if getattr(self, "h", None):
            self.lib.pislam_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def brief_table() -> np.ndarray:
    p = load().pislam_brief_table()
    return np.ctypeslib.as_array(p, shape=(30, 256, 4)).copy()
