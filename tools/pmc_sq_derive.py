#!/usr/bin/env python3
"""Sums the SQ / GRBM counters of tools/pmc_sq.sh per kernel (dominant kernels only) and joins them with the kernels' durations."""
import collections
import csv
import glob
import json
import os
import sys

R = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("syrk_segs_kernel<4>", "syrk_segs_kernel<2>", "predict_gauss", "kbuild_slab", "bulk_fwd", "bulk_bwd")


def short(nm):
    for k in KEEP:
        if k in nm:
            return k
    return None


derived = {}
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "pmcsq_[0-9]"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(float)
    for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            if k:
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    print("== %s" % os.path.basename(d))
    for k in agg:
        print("%s: %d dispatches, %.3f ms in all" % (k, len(disp[k]), dur[k] * 1e3))
        for c, v in sorted(agg[k].items()):
            extra = ""
            if c == "GRBM_GUI_ACTIVE" and dur[k] > 0:
                extra = "  -> effective clock %.3f GHz (sum over 8 XCDs / 8 / time)" % (v / 8 / dur[k] / 1e9)
            print("    %-34s %.6e%s" % (c, v, extra))
    k = "syrk_segs_kernel<4>"
    if k in agg and dur[k] > 0 and "GRBM_GUI_ACTIVE" in agg[k]:
        cyc = agg[k]["GRBM_GUI_ACTIVE"] / 8.0                      # shader cycles the kernel was resident (per XCD)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k]:
            derived["effective_clock_ghz"] = cyc / dur[k] / 1e9
            derived["mfma_busy_frac"] = agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)      # 256 CUs x 4 SIMDs
            w = agg[k]["SQ_WAVE_CYCLES"]
            derived["wave_cycles_split"] = {"issue_stall": agg[k]["SQ_WAIT_INST_ANY"] / w, "parked_waitcnt_or_barrier": agg[k]["SQ_WAIT_ANY"] / w,
                                            "issuing": agg[k]["SQ_ACTIVE_INST_ANY"] / w, "issue_stall_on_lds": agg[k]["SQ_WAIT_INST_LDS"] / w}
            derived["launches"] = len(disp[k])
        if "SQ_LDS_IDX_ACTIVE" in agg[k]:
            derived["lds_busy_frac"] = agg[k]["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
            derived["lds_bank_conflict_over_active"] = agg[k]["SQ_LDS_BANK_CONFLICT"] / agg[k]["SQ_LDS_IDX_ACTIVE"]
if derived:
    sys.path.insert(0, R)
    from bench import _csrc_digest  # noqa: E402
    derived["csrc_digest"] = _csrc_digest()
    derived["kernel"] = "syrk_segs_kernel<4>"
    derived["note"] = ("tools/pmc_sq.sh: rocprofv3 --pmc passes (kernel-trace only) over tools/quick_perf.py 65536; effective clock = "
                       "GRBM_GUI_ACTIVE / 8 XCDs / the kernel's time; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES (64 per "
                       "v_mfma_f64_16x16x4) / (cycles x 1024 SIMDs); profiled passes clock ~2 % lower than plain runs")
    json.dump(derived, open(os.path.join(R, "gpurun_out", "pmc_sq.json"), "w"), indent=1)
    print("derived:", json.dumps(derived))
