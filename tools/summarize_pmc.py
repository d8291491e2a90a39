#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of tools/profile_round.sh to the summaries kept under profiles/:
python tools/summarize_pmc.py gpurun_out/prof_<tag> profiles/r01_<tag>   (per-launch means over the timed launches of
the tick kernel; FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 reports them on gfx950)"""
import csv
import glob
import json
import os
import shutil
import sys


def rows(d, pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def counters(d):
    acc = {}
    for r in rows(d, "*counter_collection.csv"):
        if not r["Kernel_Name"].startswith("qpc_tick_kernel"):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return acc


def main():
    src, dst = sys.argv[1], sys.argv[2]
    stats = [r for r in rows(os.path.join(src, "stats"), "*kernel_stats.csv")]
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, dst + "_kernel_stats.csv")
    traffic = {}
    for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("FETCH_SIZE", "fetch_rot"), ("WRITE_SIZE", "write_rot")):
        v = counters(os.path.join(src, sub)).get(name, [])
        v = v[3:] if len(v) > 3 else v  # drop the warm-up launches
        rot = sub.endswith("_rot")  # (the passes with the inputs rotated through 9 distinct copies: no reuse between launches)
        if not rot:
            for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
                shutil.copy(f, f"{dst}_pmc_{name.lower()}.csv")
        if v or not rot:
            traffic[name + ("_inputs_rotated" if rot else "")] = {"per_launch_KiB": sum(v) / max(len(v), 1), "launches": len(v), "min": min(v, default=0), "max": max(v, default=0)}
    # identity of the kernel these counters belong to: bench.py quotes the traffic only for the kernel it was measured on
    try:
        bj = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        traffic["kernel_fingerprint"] = bj["config"].get("kernel_fingerprint")
        traffic["qps_per_launch"] = bj["config"]["batch_per_gpu"]
    except (OSError, ValueError, KeyError, IndexError):
        traffic["kernel_fingerprint"] = None
    json.dump(traffic, open(dst + "_traffic_summary.json", "w"), indent=1)
    sq = {}
    for sub in ("sq_a", "sq_b", "sq_c"):
        for k, v in counters(os.path.join(src, sub)).items():
            v = v[3:] if len(v) > 3 else v
            sq[k] = sum(v) / max(len(v), 1)
    if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU"):
        # rocprofiler's VALUUtilization: active lanes per vector instruction / 64 (lanes switched off by the exec mask do not
        # count; lanes that compute a value nobody uses do)
        sq["valu_lane_utilisation"] = sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])
    tick = [r for r in stats if r["Name"].startswith("qpc_tick_kernel")]
    json.dump({"kernel": tick[0]["Name"] if tick else None, "qps_per_launch": 32768,
               "kernel_avg_ms_under_rocprof": float(tick[0]["AverageNs"]) / 1e6 if tick else None, "per_launch": sq},
              open(dst + "_pmc_summary.json", "w"), indent=1)
    for f in glob.glob(os.path.join(src, "l0", "**", "*kernel_stats.csv"), recursive=True):  # the L0 kernels (tools/bench_latency.py --l0)
        shutil.copy(f, dst + "_l0_kernel_stats.csv")
    for name in ("bench_full.json", "bench_under_rocprof.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), f"{dst}_{name}")
    print(json.dumps({"traffic": traffic, "kernel_stats": [(r["Name"][:40], r["Calls"], r["AverageNs"]) for r in stats[:3]]}, indent=1))


if __name__ == "__main__":
    main()
