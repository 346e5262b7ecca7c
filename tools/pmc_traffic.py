#!/usr/bin/env python3
"""Writes one workload entry of profiles/pmc_traffic.json from a tools/pmc.sh run of bench.py.
usage: pmc_traffic.py <pmc dir> <source label>      (prints the JSON entry; paste / merge into profiles/pmc_traffic.json)

Per kernel, mean over the dispatches of a pass:
  hbm_bytes      = 2 x FETCH_SIZE + WRITE_SIZE (KiB x 1024): the gfx950 correction of the guide's HBM section (a 128-B read
                   request is tallied as 64 B).  Calibrated here on build_records_kernel, a coalesced u16 stream with a known
                   byte count (K x 640 x 480 x 4 B): raw FETCH_SIZE = 0.501 of it.  For sparse 4-8 B gathers the doubled figure
                   is an upper bound (whole 128-B lines); hbm_bytes_raw is the untouched FETCH_SIZE + WRITE_SIZE
  pairs_per_launch = surfels x average keyframes per launch, from the bench line printed by the same pass
  valu_wave_insts_per_pair = SQ_INSTS_VALU / pairs_per_launch   (wave instructions: x 64 for per-thread instructions)
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash   # noqa: E402  (the entry is stamped with the kernel sources it was measured on)

root, source = sys.argv[1], sys.argv[2]
bench = None
for log in sorted(glob.glob(root + "/pass*.log")):
    for line in open(log, errors="replace"):
        if line.startswith('{"metric"'):
            bench = json.loads(line)
assert bench is not None, "no bench line in the pass logs"
S, K = bench["config"]["surfels_per_gpu"], bench["config"]["keyframes"]
photometric = "photometric+geometric" in bench["config"]["workload"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bslam::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
active = bench["config"]["active_surfel_fraction"] * S
units = {"pose_accumulate_kernel": S * bench["roofline"]["keyframes_per_launch"], "pcg_step1_kernel": S * K, "pcg_init_kernel": S * K}
kernels = {}
for name, m in mean.items():
    if "FETCH_SIZE" not in m:
        continue
    short = name.split("<")[0]
    e = {"hbm_bytes": int(2048 * m["FETCH_SIZE"] + 1024 * m.get("WRITE_SIZE", 0)), "hbm_bytes_raw": int(1024 * (m["FETCH_SIZE"] + m.get("WRITE_SIZE", 0))),
         "dispatches": len(acc[name]["FETCH_SIZE"])}
    if "TCC_HIT_sum" in m:
        e["l2_hit"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
    if short in units:
        e["pairs_per_launch"] = units[short]
        if "SQ_INSTS_VALU" in m:
            e["valu_wave_insts_per_pair"] = m["SQ_INSTS_VALU"] / units[short]
    elif "SQ_INSTS_VALU" in m:
        e["valu_wave_insts"] = m["SQ_INSTS_VALU"]
    kernels[short if short in units else name] = e   # the dominant kernel under its plain name (one instantiation per workload)
# the geometry iteration of one step = every dispatch of the geometry kernels of that step
geo = [n for n in kernels if n.startswith("geometry_")]
if geo:
    steps = bench["steps"] + bench["warmup"]
    tot = sum(kernels[n]["hbm_bytes"] * kernels[n]["dispatches"] for n in geo) / steps
    raw = sum(kernels[n]["hbm_bytes_raw"] * kernels[n]["dispatches"] for n in geo) / steps
    kernels["geometry_kernel"] = {"hbm_bytes": int(tot), "hbm_bytes_raw": int(raw), "per": "step (all launches of the normals and position passes)",
                                  "pairs_per_step": 2 * K * active}
print(json.dumps({"keyframes": K, "surfels_per_gpu": S, "photometric": photometric, "scene": bench["config"].get("scene", "dense"), "csrc_hash": csrc_hash(),
                  "source": source, "kernels": kernels}, indent=1))
