#!/usr/bin/env python3
"""Instructions per pair iteration of the pair-binning kernels (csrc/kk.hip: kk_pairs_kernel<Log> / <TwoD>), counted in the
ISA hipcc emits for gfx950: the pair loop's body, all paths (a wave runs every path some lane takes) and without the
division fall-backs of bin_of (taken only next to a bin edge: one v_rcp_f64 + ~13 instructions each).  bench.py's roofline_kk_*
price the kernels against the VALU issue rate with the second figure (a wave64 VALU instruction occupies its SIMD for at
least 4 clocks: tools/probes/issue_probe.hip measures 4.5 for 64-bit ones).
usage: python tools/kk_isa_count.py"""
import os
import re
import subprocess
import tempfile

here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "treegp_amd", "csrc", "kk.hip")
out = os.path.join(tempfile.mkdtemp(), "kk.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src],
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for name in re.findall(r"^(_ZN[^\n:]*kk_pairs_kernel[^\n:]*):", s, re.M):
    body = s[s.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")]
    lines = [l.strip() for l in body.split("\n")]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    back = []
    for i, l in enumerate(lines):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            back.append((labels[m.group(1)], i))
    # the pair loop: the widest loop whose body holds the LDS atomics (ds_add_f64) and no global load
    cand = [(a, b) for a, b in back if any("ds_add_f64" in l for l in lines[a:b + 1]) and not any(l.startswith("global_load") for l in lines[a:b + 1])]
    a = min(x for x, _ in cand)
    b = max(y for x, y in cand if x == a)
    seg = [l for l in lines[a:b + 1] if l and not l.startswith((".", ";"))]
    valu = [l for l in seg if l.startswith("v_")]
    rcp = sum(1 for l in valu if l.startswith("v_rcp_f64"))
    atom = sum(1 for l in seg if l.startswith("ds_add_f64"))
    print("%s: pair loop = %d instructions, %d VALU (%d of them 64-bit), %d LDS atomics, %d division fall-backs -> ~%d VALU on the common path"
          % ("TwoD" if "ILb1" in name else "Log ", len(seg), len(valu), sum(1 for l in valu if re.search(r"_f64|_b64|_u64|_i64", l)), atom,
             rcp, len(valu) - 14 * rcp))
