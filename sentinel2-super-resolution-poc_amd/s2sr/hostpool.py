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


def workers() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 4
    return max(2, min(int(os.environ.get("S2SR_HOST_THREADS", "32")), n))


def pool() -> ThreadPoolExecutor:
    global _POOL, _PID
    with _LOCK:
        if _POOL is None or _PID != os.getpid():      # a forked child starts its own
            _POOL = ThreadPoolExecutor(max_workers=workers(), thread_name_prefix="s2sr-host")
            _PID = os.getpid()
        return _POOL
