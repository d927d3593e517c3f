#!/usr/bin/env python3
"""Sweep of the launch-shape tunables over BASELINE's sizes and the two extreme scenes (GPU box): frames in flight,
the fill workgroups riding with the first launches (PAR_TUNE_FILL_WGS) and the hash build's share of the fill
(PAR_TUNE_FILL_BUILD_PCT). Every cell is the C++ host loop (par_pipeline, one submitting thread per slot), median of
three runs. Writes a JSON table (profiles/r02_sweep.json is a copy of one run).
usage: sweep.py [out.json]"""
import json, os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pixel-art-raytracer_amd", "lib", "par_pipeline")
SCENES = [("512^2/64", ["--size", "512", "--prims", "64"], 6000), ("1024^2/512", ["--size", "1024", "--prims", "512"], 6000),
          ("2048^2/256", ["--size", "2048", "--prims", "256"], 6000), ("4096^2/1024", ["--size", "4096", "--prims", "1024"], 4000),
          ("4096^2 floor", ["--scene", "floor", "--size", "4096"], 300), ("480x320 graybox", ["--scene", "graybox"], 6000)]

def run(args, frames, inflight, env):
    v = []
    for _ in range(3):
        p = subprocess.run([EXE, "--frames", str(frames), "--inflight", str(inflight), "--threads", str(inflight)] + args,
                           capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        v.append(json.loads(p.stdout.splitlines()[0])["us_per_frame"])
    return round(statistics.median(v), 2)

out = {"unit": "us per frame, median of 3 runs of par_pipeline", "cells": []}
for name, args, frames in SCENES:
    for inflight in (1, 2, 3, 4, 6):
        out["cells"].append({"scene": name, "inflight": inflight, "fill_wgs": 64, "fill_build_pct": 40,
                             "us_per_frame": run(args, frames, inflight, {})})
        print(out["cells"][-1], flush=True)
    for wgs in (16, 32, 128, 256):
        out["cells"].append({"scene": name, "inflight": 4, "fill_wgs": wgs, "fill_build_pct": 40,
                             "us_per_frame": run(args, frames, 4, {"PAR_TUNE_FILL_WGS": str(wgs)})})
        print(out["cells"][-1], flush=True)
    for pct in (0, 20, 60, 100):
        out["cells"].append({"scene": name, "inflight": 4, "fill_wgs": 64, "fill_build_pct": pct,
                             "us_per_frame": run(args, frames, 4, {"PAR_TUNE_FILL_BUILD_PCT": str(pct)})})
        print(out["cells"][-1], flush=True)
best = {}
for c in out["cells"]:
    k = c["scene"]
    if k not in best or c["us_per_frame"] < best[k]["us_per_frame"]:
        best[k] = c
out["best_per_scene"] = best
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
