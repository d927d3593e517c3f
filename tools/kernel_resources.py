#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels in par_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py   (no GPU needed)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(ROOT, "pixel-art-raytracer_amd", "csrc"), "asm"],
                     capture_output=True, text=True)
text = out.stdout + out.stderr
cur = None
rows = {}
for line in text.splitlines():
    m = re.search(r"remark: .*?: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        p = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = p.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
print(f"{'kernel':34s} {'SGPR':>5s} {'VGPR':>5s} {'occ':>4s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'LDS':>6s}")
for k, r in rows.items():
    print(f"{k:34s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs','?'):>5s} {r.get('Occupancy','?'):>4s} "
          f"{r.get('SGPRs Spill','?'):>6s} {r.get('VGPRs Spill','?'):>6s} {r.get('ScratchSize','?'):>7s} {r.get('LDS Size','?'):>6s}")
