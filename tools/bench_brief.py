#!/usr/bin/env python3
"""Prints the key figures of a bench.py JSON line (file argument), one row per block."""
import json
import sys

j = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])


def brief(d, name):
    c, r = d["config"], d["roofline"]
    g = r.get("geometry_kernel") or {}
    print(f"{name:10s} {d['value']:.3e} pairs/s  {d['ms_per_step']:8.2f} ms/step  inb {c['in_bounds_pair_fraction']:.3f}  culled {c.get('culled_pair_fraction')}"
          f"  gn/step {c['gn_iterations_per_step']:.0f}  conv {c['keyframes_converged_fraction']:.2f}\n"
          f"           pose {r['avg_launch_us']:.0f} us x{r['launches']} ({r['keyframes_per_launch']:.1f} kf/launch) frac {r['frac']:.3f} | geometry {g.get('us_per_step', 0):.0f} us frac {g.get('frac', 0):.3f}"
          f" | activation {r.get('activation_kernel', {}).get('avg_launch_us', 0):.0f} us | reduce+solve {r.get('pose_reduce_solve_kernel', {}).get('avg_launch_us', 0):.0f} us"
          f" | exchange {c.get('exchange_timing')}")


if "roofline" in j:
    brief(j, "headline")
for k in ("secondary", "trajectory", "survey", "strong"):
    if k in j.get("config", {}):
        brief(j["config"][k], k)
for k, v in (j.get("pcg") or {}).items():
    s1 = v.get("pcg_step1_kernel", {})
    print(f"pcg {k}: {v['ms_per_ba_iteration']:.2f} ms / BA iteration, {v['inner_steps_per_iteration']} CG steps, step1 {s1.get('avg_launch_us', 0):.0f} us frac {s1.get('frac', 0):.3f}, "
          f"init {v.get('pcg_init_kernel', {}).get('avg_launch_us', 0):.0f} us")
if "cpu_baseline" in j:
    print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
print("n_gpus", j.get("n_gpus"), "exchange:", j.get("config", {}).get("exchange"))
