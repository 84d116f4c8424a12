#!/usr/bin/env python3
"""Turns the counter CSVs of tools/pmc_bench.sh into gpurun_out/pmc_bench_n65536.json: raw sums per kernel, the
calibration factors measured on this box (tools/probes/fetch_calib.hip, known byte counts), corrected HBM-side bytes per
launch, and the digest of the HIP sources they were collected on (bench.py quotes `traffic` only for matching sources)."""
import collections
import csv
import glob
import json
import os
import sys

R = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from bench import _csrc_digest  # noqa: E402

KIB = 1024.0                  # FETCH_SIZE / WRITE_SIZE are reported in KiB
CAL_BYTES = float(4 << 30)    # what every calibration kernel moves


def sums(pattern, classify):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for fn in glob.glob(os.path.join(R, "gpurun_out", pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = classify(r["Kernel_Name"])
            if k is None:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    return {k: {c: {"sum": v, "dispatches": len(disp[(k, c)])} for c, v in agg[k].items()} for k in agg}


def cal_name(nm):
    for k in ("read16", "read8t", "read8", "write16", "write8"):
        if nm.startswith(k + "(") or nm.startswith("void " + k) or nm.split("(")[0].strip() == k:
            return k
    return None


def bench_name(nm):
    if "syrk_segs_kernel<4>" in nm:
        return "syrk_segs_kernel<4>"
    if "kbuild_lower" in nm or "kbuild_slab" in nm:
        return "kbuild_lower_kernel"
    return None


cal = sums("pmccal_*", cal_name)
factors = {}
for k, v in cal.items():
    c = "FETCH_SIZE" if k.startswith("read") else "WRITE_SIZE"
    if c in v and v[c]["sum"] > 0:
        factors[k] = CAL_BYTES * v[c]["dispatches"] / (v[c]["sum"] * KIB)      # true bytes / reported bytes
out = sums("pmcb_*", bench_name)
der = {"csrc_digest": _csrc_digest(), "calibration_true_over_reported": factors}
sy = out.get("syrk_segs_kernel<4>", {})
if "FETCH_SIZE" in sy and "WRITE_SIZE" in sy:
    L = sy["FETCH_SIZE"]["dispatches"]
    f16 = factors.get("read16", 2.0)
    fetch_raw = sy["FETCH_SIZE"]["sum"] * KIB
    write = sy["WRITE_SIZE"]["sum"] * KIB * factors.get("write8", 1.0)
    # the kernel's reads are 16-B/lane operand loads (the bulk) plus the 8-B/lane accumulator-shaped C preload; the C
    # part equals the bytes written (every updated element is read once and written once), so it is priced with the
    # read8t factor and the remainder with the read16 factor
    f8t = factors.get("read8t", 1.0)
    c_read_reported = write / f8t
    fetch = write + max(fetch_raw - c_read_reported, 0.0) * f16
    der.update(syrk_launches=L, syrk_fetch_bytes_reported=fetch_raw, syrk_fetch_bytes=fetch, syrk_write_bytes=write,
               syrk_hbm_bytes_per_launch=(fetch + write) / L, syrk_algorithmic_C_bytes_per_launch=2.0 * write / L,
               # the profiled run is ONE step; under counter collection the library hands over by events, so the update runs as
               # two launches per group where the timed run fuses them: bench.py divides the step's total by ITS launch count
               syrk_hbm_bytes_per_step=fetch + write)
    if "TCC_HIT_sum" in sy:
        der["syrk_L2_hit_rate"] = sy["TCC_HIT_sum"]["sum"] / (sy["TCC_HIT_sum"]["sum"] + sy["TCC_MISS_sum"]["sum"])
kb = out.get("kbuild_lower_kernel", {})
if "WRITE_SIZE" in kb:
    der["kbuild_write_bytes"] = kb["WRITE_SIZE"]["sum"] * KIB * factors.get("write16", 1.0) / kb["WRITE_SIZE"]["dispatches"]
der["note"] = ("one solve at N=65536 (python3 bench.py --steps 1 --warmup 0 --cpu-sample 0), separate rocprofv3 --pmc passes "
               "(FETCH_SIZE / WRITE_SIZE / TCC hit+miss) with --kernel-trace only; counters in KiB; corrected with factors measured "
               "on the same box by tools/probes/fetch_calib.hip (true/reported for 4 GiB moved once per kernel); Infinity-Cache "
               "hits are counted by these fabric-side counters, so `traffic` is an upper bound on HBM bytes")
out["_calibration"] = cal
out["_derived"] = der
json.dump(out, open(os.path.join(R, "gpurun_out", "pmc_bench_n65536.json"), "w"), indent=1)
print(json.dumps(der, indent=1))
