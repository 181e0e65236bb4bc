"""One thread pool for the host codecs (TIFF strip decode / LZW encode, PNG band deflate): created at first use and kept.
A pool per call cost a thread start per worker and call (tens of them per job), and the GeoTIFF and the PNG writer of one job,
which run side by side, each started 16 threads on a 16-CPU quota.  Tasks submitted here must not wait on other tasks of this
pool (the codecs' band / strip tasks are leaves)."""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor

_LOCK = threading.Lock()
_POOL = None
_PID = None


def env_int(name: str, default: int) -> int:
    """A positive integer from the environment; anything else (unset, malformed, <= 0) gives the default -- what the native side's
    getenv / atoi handling does, so a typo in a knob never surfaces as a codec error in the middle of a job."""
    try:
        v = int(os.environ.get(name, ""))
    except ValueError:
        return default
    return v if v > 0 else default


def usable_cpus() -> int:
    """CPUs this process may really use: its affinity mask, capped by the cgroup quota (cpu.max: a container may see 256 logical
    CPUs and be allowed 16 of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 4
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return n


def workers() -> int:
    return max(2, min(env_int("S2SR_HOST_THREADS", 32), usable_cpus()))


def pool() -> ThreadPoolExecutor:
    global _POOL, _PID
    with _LOCK:
        if _POOL is None or _PID != os.getpid():      # a forked child starts its own
            _POOL = ThreadPoolExecutor(max_workers=workers(), thread_name_prefix="s2sr-host")
            _PID = os.getpid()
        return _POOL
