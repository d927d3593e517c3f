#!/usr/bin/env python3
"""GPU box: what the tile gather's kernels cost on one GPU (DESIGN section 6's table): par_tiles_pack of a rank's run,
par_background_fill of the whole frame and par_tiles_unpack of every run on the assembling rank, for the headline
scene split over N = 1, 2, 4, 8 ranks (the sends themselves need the other GPUs). usage: gather_cost.py [size] [prims]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
par = importlib.import_module("pixel-art-raytracer_amd")
T = importlib.import_module("pixel-art-raytracer_amd.types")
sh = importlib.import_module("pixel-art-raytracer_amd.sharding")
pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    prims = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    p = T.default_params(size, size, size)
    aabbs, light = par.scene_synthetic(prims, size, size, size, 12345)
    sprite = par.tile_floor()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    out = {"size": size, "primitives": prims, "ranks": {}}
    for world in (1, 2, 4, 8):
        root = sh.TileGather(p, aabbs, dev, world=world, rank=0)
        last = sh.TileGather(p, aabbs, dev, world=world, rank=world - 1, dst=0) if world > 1 else root
        block, packed = last.block_buffer(), last.packed_buffer()
        e = {"tiles": len(root.tiles), "tiles_per_rank": root.counts,
             "bytes_to_rank0": sum(root.bytes_sent(r) for r in range(world)),
             "largest_send_bytes": max([root.bytes_sent(r) for r in range(world)] + [0]),
             "pack_us_last_rank": round(timed(lambda: last.pack(block, packed, stream)), 2),
             "background_fill_us": round(timed(lambda: par.background_fill(p, root.frame.data_ptr(), p.height, stream)), 2),
             "unpack_all_us": round(timed(lambda: par.tiles_unpack(p, root.d_tiles.data_ptr(), len(root.tiles),
                                                                  root.inbox.data_ptr(), root.frame.data_ptr(), stream)), 2),
             "assemble_us": round(timed(lambda: root.assemble(stream)), 2)}
        # in place: rank 0 renders its block into the frame and writes the other rows once (par_tiles_assemble)
        inp = sh.TileGather(p, aabbs, dev, world=world, rank=0, in_place=True)
        e["assemble_in_place_us"] = round(timed(lambda: inp.assemble(stream)), 2)
        # a rank's render of its own block with four frames in flight and nothing else on the GPU (first and last
        # rank: the blocks differ in what they show)
        per = {}
        for r in sorted({0, world - 1}):
            pipe = pipeline.FramePipeline(p, aabbs, sprite, light, depth=4, device=0, rows=root.blocks[r],
                                          planes=("fb", "palidx"))
            try:
                pipe.submit_many(0, 400)
                pipe.synchronize()
                ms = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    pipe.submit_many(0, 1000)
                    pipe.synchronize()
                    ms.append((time.perf_counter() - t0) / 1000 * 1e6)
                per[str(r)] = round(float(np.median(ms)), 2)
            finally:
                pipe.close()
        e["render_block_us_per_frame"] = per
        out["ranks"][str(world)] = e
    print(json.dumps(out))


if __name__ == "__main__":
    main()
