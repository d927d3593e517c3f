#!/usr/bin/env python3
"""GPU box: par_tiles_assemble of 7/8 of the headline frame for different grid sizes (PAR_TUNE_ASSEMBLE_WGS is read
once per process, so each value runs in a child)."""
import os
import subprocess
import sys

CHILD = r'''
import importlib, sys, torch
sys.path.insert(0, %r)
par = importlib.import_module("pixel-art-raytracer_amd")
T = importlib.import_module("pixel-art-raytracer_amd.types")
sh = importlib.import_module("pixel-art-raytracer_amd.sharding")
p = T.default_params(4096, 4096, 4096)
aabbs, light = par.scene_synthetic(1024, 4096, 4096, 4096, 12345)
g = sh.TileGather(p, aabbs, torch.device("cuda:0"), world=8, rank=0, in_place=True)
st = torch.cuda.current_stream().cuda_stream
for _ in range(10): g.assemble(st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): g.assemble(st)
b.record(); torch.cuda.synchronize()
print(round(a.elapsed_time(b) * 10, 2))
'''
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for wgs in sys.argv[1:] or ["512", "1024", "2048", "4096", "8192", "16384", "65536"]:
    out = subprocess.run([sys.executable, "-c", CHILD % root], env=dict(os.environ, PAR_TUNE_ASSEMBLE_WGS=wgs),
                         capture_output=True, text=True)
    print(wgs, out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else "", flush=True)
