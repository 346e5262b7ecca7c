#!/usr/bin/env python3
"""Merges workload entries printed by tools/pmc_traffic.py into profiles/pmc_traffic.json (an entry replaces the one of the same
workload: keyframes, surfels, residuals, scene).  Only the library's own kernels are kept.
usage: tools/pmc_merge.py <entry.json> [<entry.json> ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
doc = json.load(open(path))


def key(w):
    return (w.get("keyframes"), w.get("surfels_per_gpu"), bool(w.get("photometric")), w.get("scene", "dense"))


for f in sys.argv[1:]:
    e = json.load(open(f))
    e["kernels"] = {k: v for k, v in e["kernels"].items() if not k.startswith(("__amd", "at::", "rocprim", "void at::"))}
    doc["workloads"] = [w for w in doc["workloads"] if key(w) != key(e)] + [e]
doc["workloads"].sort(key=lambda w: (w.get("scene", "dense"), -w.get("keyframes", 0), bool(w.get("photometric"))))
json.dump(doc, open(path, "w"), indent=1)
print(len(doc["workloads"]), "workloads:", [key(w) + (w.get("csrc_hash"),) for w in doc["workloads"]])
