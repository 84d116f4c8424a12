#!/usr/bin/env python3
"""Prints the parts of a bench.py JSON line a round looks at first.  usage: show_bench.py <file with the line>"""
import json
import sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("ms_per_step %.1f  value %.3e %s" % (o["ms_per_step"], o["value"], o["unit"]))
for k in ("roofline", "roofline_kbuild", "roofline_trsv", "roofline_predict"):
    if k in o:
        r = o[k]
        print("%-18s %-6s achieved %.4g of %.4g %s = %.3f" % (k, r["bound"], r["achieved"], r["peak"], r["unit"], r["frac"]))
for c in o.get("configs_measured", []):
    print({k: (v if not isinstance(v, dict) else "...") for k, v in c.items()})
    for k in ("roofline_kk", "roofline_predict_vk", "api_route_ms"):
        if k in c:
            print("   ", k, c[k])
cb = o.get("cpu_baseline")
if isinstance(cb, dict):
    print("cpu_baseline value %.3e %s cores %s" % (cb["value"], cb["unit"], cb["cores"]))
    print("   ", cb.get("configs", {}).get("configs[2]", {}).get("pair_binning"))
