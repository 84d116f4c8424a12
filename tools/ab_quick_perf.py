#!/usr/bin/env python3
"""A/B aid: run tools/quick_perf.py against an older libtgp build (TGP_LIB_PATH) that may lack newer symbols."""
import ctypes
import os
import runpy
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib  # noqa: E402

raw = ctypes.CDLL(os.environ.get("TGP_LIB_PATH", _lib.LIB_PATH))
for name in list(_lib.SIGNATURES):
    if not hasattr(raw, name):
        del _lib.SIGNATURES[name]
runpy.run_path(__file__.rsplit("/", 1)[0] + "/quick_perf.py", run_name="__main__")
