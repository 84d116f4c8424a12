#!/usr/bin/env python3
"""The trailing update alone, back to back on random data (same conditions as tools/vendor_dgemm.py's
sustained GEMM), with rocm-smi clock/power samples alongside.  Default: the depth-512 kernel; with
TGP_DEBUG_SEGS=1 in the environment the depth-1024 kernel of the groups-of-four schedule."""
import ctypes as C
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402

samples, stop = [], False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            f = out.strip().splitlines()[1].split(",")
            samples.append((time.perf_counter(), f[5], f[9]))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.1)


def main():
    global stop
    Np = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
    lib = _lib.load_library()
    ctx = _lib.get_ctx()
    ne = lib.tgp_panel_elems(Np)
    rng = np.random.default_rng(0)
    host = (scale * rng.standard_normal(min(ne, 1 << 26))).astype(np.float64)
    buf = ops.DeviceBuffer(ctx, ne * 8)
    off = 0
    while off < ne:                                      # tile the random block over the whole matrix
        cnt = min(len(host), ne - off)
        _lib.check(ctx, lib.tgp_h2d(ctx, C.c_void_p(buf.ptr.value + off * 8), host.ctypes.data_as(C.c_void_p), cnt * 8), "h2d")
        off += cnt
    th = threading.Thread(target=poll)
    th.start()
    ms, fl = C.c_double(), C.c_double()
    for chunk in range(5):
        _lib.check(ctx, lib.tgp_debug_syrk_loop(ctx, buf.ptr, Np, reps, C.byref(ms), C.byref(fl)), "syrk_loop")
        print("chunk %d: Np=%d  %.3f ms/launch  %.2f TFLOP/s" % (chunk, Np, ms.value, fl.value / ms.value / 1e9), flush=True)
    stop = True
    th.join()
    t0 = samples[0][0] if samples else 0
    print(" ".join("%.1fs:%s/%sW" % (t - t0, c.strip("()"), p) for t, c, p in samples[::3]))


if __name__ == "__main__":
    main()
