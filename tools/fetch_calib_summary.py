#!/usr/bin/env python3
"""Reduce tools/fetch_calib.sh's passes: per calibration kernel counted / known bytes, the raw request counters where this ROCm
collects them, and the tick's FETCH_SIZE restated with the factor of the pattern it reads in.
python tools/fetch_calib_summary.py gpurun_out/calib_<tag> out.json"""
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            acc.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v[1:]) / max(len(v) - 1, 1) if len(v) > 1 else v[0] for c, v in cs.items()} for k, cs in acc.items()}  # (first launch dropped)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    known = {}
    for line in open(os.path.join(src, "known.jsonl")):
        line = line.strip()
        if line.startswith("{"):
            j = json.loads(line)
            known[j["kernel"]] = j
    fetch, raw, raw2 = per_kernel(os.path.join(src, "fetch")), per_kernel(os.path.join(src, "raw")), per_kernel(os.path.join(src, "raw2"))
    out = {"unit": "FETCH_SIZE in KiB as rocprofv3 reports it on gfx950; factor = counted bytes / known bytes", "kernels": {}}
    for k, j in known.items():
        e = dict(j)
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            e["fetch_size_KiB"] = fetch[k]["FETCH_SIZE"]
            e["factor"] = fetch[k]["FETCH_SIZE"] * 1024.0 / j["known_bytes"]
        for src_ in (raw, raw2):
            for c, v in src_.get(k, {}).items():
                e[c] = v
                e[c + "_per_known_128B"] = v / (j["known_bytes"] / 128.0)
        out["kernels"][k] = e
    tick = {}
    for sub in ("tick_fetch", "tick_raw", "tick_raw2"):
        for k, cs in per_kernel(os.path.join(src, sub)).items():
            if k.startswith("qpc_tick_kernel"):
                tick.update(cs)
    out["tick_inputs_rotated"] = tick
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
