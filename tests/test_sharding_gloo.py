"""The N > 1 path on CPU: world_size 2 and 3 over gloo. Row blocks rendered independently and gathered to rank 0
must be byte-identical to the whole frame — for even splits (blocks land straight in the frame) and
for uneven ones (padded staging + unpack)."""
import importlib
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_row_blocks_partition():
    """Blocks are contiguous, cover the frame, are cut at bin rows and hold bin-row counts that differ by
    at most one; C (par_row_block) and Python agree because Python calls C."""
    sharding = importlib.import_module("pixel-art-raytracer_amd.sharding")
    for b in (40, 16, 64):
        for h in (1, 7, 320, 4096, 4099, 2048):
            for n in (1, 2, 3, 4, 8):
                blocks = [sharding.row_block(r, n, h, b) for r in range(n)]
                assert blocks[0][0] == 0 and blocks[-1][1] == h
                assert all(blocks[i][1] == blocks[i + 1][0] for i in range(n - 1))
                assert all(beg % b == 0 for beg, _ in blocks)
                bins = [-(-(e - beg) // b) for beg, e in blocks]
                assert max(bins) - min(bins) <= 1
                assert sharding.max_block_rows(n, h, b) == max(e - beg for beg, e in blocks)
    # BASELINE's sizes at bin 40: 4096 rows are 103 bin rows (the last one 16 rows)
    assert [sharding.row_block(r, 8, 4096) for r in range(8)] == [(0, 480), (480, 1000), (1000, 1520), (1520, 2040),
                                                                   (2040, 2560), (2560, 3080), (3080, 3600), (3600, 4096)]


@pytest.mark.parametrize("world,height,mode", [(2, 160, "blocks"), (3, 200, "blocks"), (2, 320, "blocks"),
                                               (2, 320, "tiles"), (3, 200, "tiles"), (2, 320, "tiles_in_place"),
                                               (3, 200, "tiles_in_place")])
def test_gather_assembles_the_frame(tmp_path, world, height, mode):
    """mode "blocks": whole row blocks gathered (FrameGather); "tiles": only the tiles that can show a primitive
    travel and the root writes the background itself (TileGather); "tiles_in_place": the same with the root's own
    block produced straight into the assembled frame."""
    out = tmp_path / "result.txt"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(height), "240", str(out), mode]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert out.read_text() == "ok"


def test_scene_tiles_cover_every_covered_pixel():
    """par_scene_tiles (host arithmetic): sorted by bin row, and a superset of the tiles in which the oracle covers a
    pixel -- what makes it safe to let only those tiles travel."""
    import numpy as np
    par = importlib.import_module("pixel-art-raytracer_amd")
    from oracle.oracle import Oracle
    T = par.types
    o = Oracle()
    for (w, h, l, n, seed, b) in [(240, 200, 200, 60, 5, 40), (333, 170, 90, 150, 9, 16), (128, 128, 128, 3, 1, 64)]:
        params = T.default_params(w, h, l, b)
        aabbs, light = par.scene_synthetic(n, w, h, l, seed)
        tiles = par.scene_tiles(params, aabbs)
        keys = (tiles >> 16) * 65536 + (tiles & 0xFFFF)
        assert np.all(np.diff(keys) > 0)
        pal = o.render(params, aabbs, par.tile_floor(), light, planes=("palidx",))["palidx"].reshape(h, w)
        ys, xs = np.nonzero(pal != T.PALIDX_BACKGROUND)
        covered = set(zip((xs // b).tolist(), (ys // b).tolist()))
        listed = set(zip((tiles & 0xFFFF).tolist(), (tiles >> 16).tolist()))
        assert covered <= listed
    assert len(par.scene_tiles(T.default_params(), T.make_aabbs([]))) == 0
