"""s2sr: host side of the MI355X Real-ESRGAN x4 path (weights, ctypes binding, sharding)."""
