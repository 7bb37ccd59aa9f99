#!/usr/bin/env python3
"""Runs the device-arithmetic self-tests of libvrt_selftest.so and prints what each found."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "raytracing_amd", "libvrt_selftest.so"))
for name, n in (("vrt_selftest_recip", 3), ("vrt_selftest_normalize", 4), ("vrt_selftest_unorm8", 2), ("vrt_selftest_floor_convert", 2)):
    f = getattr(lib, name)
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
    out = (C.c_ulonglong * n)()
    rc = f(0, out)
    print(name, "rc", rc, [hex(v) for v in out], flush=True)
