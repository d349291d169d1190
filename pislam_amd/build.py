"""Build libpislam_hip.so in-tree with hipcc for gfx950 (no JIT cache, no CPU fallback)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libpislam_hip.so")
# Development only (A/B runs of kernel variants on the GPU box): PISLAM_HIP_LIB names another in-tree build of the same
# sources (tools/ab_build.sh); it is loaded as it is, never rebuilt.
_LIB_OVERRIDE = os.environ.get("PISLAM_HIP_LIB")
if _LIB_OVERRIDE:
    LIB = os.path.abspath(_LIB_OVERRIDE)
_override_announced = False


def check_override() -> None:
    """Called once by capi.load(): the override must exist (it is never built) and is announced on stderr — once per
    process, not at import time (bench worker processes import this module without ever loading the library)."""
    global _override_announced
    if not _LIB_OVERRIDE or _override_announced:
        return
    if not os.path.exists(LIB):
        raise RuntimeError(f"PISLAM_HIP_LIB={_LIB_OVERRIDE}: no such library (the override is never built; unset it or run "
                           "tools/ab_build.sh)")
    import sys as _sys
    print(f"[pislam_amd] PISLAM_HIP_LIB override: loading {LIB} instead of the in-tree build", file=_sys.stderr)
    _override_announced = True
SOURCES = ["pislam_hip.hip"]


def _deps():
    """Every header / table the sources include: all of csrc/*.h and *.inc, plus the public ABI header."""
    import glob
    d = [os.path.join(CSRC, f) for f in SOURCES]
    d += sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")))
    d.append(os.path.join(os.path.dirname(PKG), "include", "pislam_hip.h"))
    return d

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden"]


def source_hash() -> str:
    """SHA-256 (16 hex digits) over the kernel sources: profile files record it, so that a number measured
    on older kernels is recognised as stale (bench.py `roofline.traffic`)."""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        if os.path.exists(d) and not d.endswith("pislam_hip.h"):
            h.update(os.path.basename(d).encode())
            h.update(open(d, "rb").read())
    return h.hexdigest()[:16]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU fallback)")


def _stale() -> bool:
    if _LIB_OVERRIDE:
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library if missing or older than its sources; returns its path.

    Safe under concurrent callers (one process per GPU under torchrun): an exclusive file lock
    serialises builders, the compiler writes to a temporary name and the result is renamed into
    place atomically, and late arrivals re-check staleness after taking the lock."""
    if _LIB_OVERRIDE or (not force and not _stale()):
        return LIB
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            tmp = LIB + f".tmp{os.getpid()}"
            cmd = [hipcc()] + FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
