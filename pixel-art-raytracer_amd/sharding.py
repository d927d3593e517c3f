"""Row-block sharding of one frame across the GPUs of a node — new design, the reference is
single-process.

Every pixel is independent in both passes of the reference (alt:277-379 and alt:703-759 read only their own texel
and the read-only spatial hash), so the frame shards by contiguous row blocks: rank r renders rows
[H*r/N, H*(r+1)/N) of the same frame. The scene and the spatial hash are replicated (shadow rays leave the row
block, so every rank needs the whole hash; it is rebuilt per GPU from 16 B per primitive). The only exchange step
is assembling the frame: one gather of the row blocks to the root over RCCL/xGMI (`torch.distributed`, backend
"nccl" on ROCm; "gloo" in the CPU tests). No reduction takes place anywhere.
"""
import torch
import torch.distributed as dist


def row_block(rank, world, height, bin_size=40):
    """Rows [begin, end) of `rank`. Blocks are contiguous, cover [0, height) exactly and are cut at multiples of the
    bin size (a bin row never straddles two ranks, so no screen column is recorded twice); the bin rows
    are dealt as evenly as they go (a rank may get none when there are more ranks than bin rows). The arithmetic
    lives in the library (par_row_block) so that the C++ and Python hosts agree."""
    from . import row_block as lib_row_block
    return lib_row_block(rank, world, height, bin_size)


def max_block_rows(world, height, bin_size=40):
    return max(row_block(r, world, height, bin_size)[1] - row_block(r, world, height, bin_size)[0]
               for r in range(world))


class _Done:
    """A completed work handle."""

    def wait(self):
        return True


class FrameGather:
    """Assembles row blocks on `dst`. With equal blocks the root receives straight into views of the final frame
    (no staging copy); otherwise blocks are padded to the largest one and unpacked on the root."""

    def __init__(self, height, row_elems, dtype, device, world=None, rank=None, dst=0, group=None, bin_size=40):
        self.group = group
        self.world = dist.get_world_size(group) if world is None else world
        self.rank = dist.get_rank(group) if rank is None else rank
        self.dst = dst
        self.height, self.row_elems = height, row_elems
        self.blocks = [row_block(r, self.world, height, bin_size) for r in range(self.world)]
        self.max_rows = max(e - b for b, e in self.blocks)
        self.equal = all(e - b == self.max_rows for b, e in self.blocks)
        self.frame = None
        self.staging = None
        if self.rank == dst:
            self.frame = torch.zeros(height * row_elems, dtype=dtype, device=device)
            if not self.equal:
                self.staging = [torch.zeros(self.max_rows * row_elems, dtype=dtype, device=device)
                                for _ in range(self.world)]

    def block_buffer(self, dtype, device):
        """A send buffer for this rank's block (padded to the largest block)."""
        return torch.zeros(self.max_rows * self.row_elems, dtype=dtype, device=device)

    def gather(self, block, async_op=False):
        """`block`: this rank's padded block buffer. Returns the work handle when async_op, else None; the assembled
        frame is `self.frame` on the root."""
        if block.is_cuda and dist.get_backend(self.group) == "gloo":
            return self._gather_via_host(block)
        glist = None
        if self.rank == self.dst:
            if self.equal:
                n = self.max_rows * self.row_elems
                glist = [self.frame[r * n:(r + 1) * n] for r in range(self.world)]
            else:
                glist = self.staging
        work = dist.gather(block, glist, dst=self.dst, group=self.group, async_op=async_op)
        if not async_op:
            self.unpack()
        return work

    def _gather_via_host(self, block):
        """Test path (gloo has no device gather): the same exchange staged through host memory, synchronously. Lets
        the N > 1 control flow of bench.py be exercised where RCCL cannot run (several ranks sharing one GPU)."""
        torch.cuda.current_stream().synchronize()
        host = block.cpu()
        glist = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
        dist.gather(host, glist, dst=self.dst, group=self.group)
        if self.rank == self.dst:
            n = self.max_rows * self.row_elems
            for r, t in enumerate(glist):
                if self.equal:
                    self.frame[r * n:(r + 1) * n].copy_(t)
                else:
                    self.staging[r].copy_(t)
        return _Done()

    def unpack(self):
        """Uneven blocks only: copy the padded staging buffers into the frame (after the gather completed)."""
        if self.rank == self.dst and not self.equal:
            for r, (b, e) in enumerate(self.blocks):
                n = (e - b) * self.row_elems
                self.frame[b * self.row_elems:b * self.row_elems + n].copy_(self.staging[r][:n])


class TileGather:
    """Assembles a sharded frame on `dst` from the TILES that can show a primitive instead of whole row blocks.

    Most of a sparse frame is the constant background ({127,127,127,0} x ambient, alt:281, 735), which the assembling
    rank writes itself; only the bin_size x bin_size screen tiles the entities reach (par_scene_tiles: the cull and bin
    ranges of alt:212-240) have to travel -- at 4096^2 with 1024 primitives about 3 000 of 10 609 tiles, 19 MB instead
    of 64 MiB. Every rank holds the whole scene, so every rank derives the same list and the same split of it (the
    list is sorted by bin row and row blocks are cut at bin rows: a rank's tiles are one contiguous run); nothing but
    pixels is exchanged. Per frame: every rank packs its run out of its block (par_tiles_pack), sends it to `dst`
    (point to point: the runs differ in length); `dst` fills the frame with the background and unpacks all runs, its
    own included (par_tiles_unpack). The scene must not change between the ranks' calls of one frame.

    `in_place`: the root renders its own block straight into its rows of the assembled frame (`root_block()`), packs
    and copies nothing of its own, and writes every OTHER row exactly once, tile or background, in one pass
    (par_tiles_assemble) -- the root is the rank with the most to do, and this takes a launch and a third of the
    bytes off it."""

    def __init__(self, params, aabbs, device, world=None, rank=None, dst=0, group=None, in_place=False):
        from . import scene_tiles
        self.group = group
        self.world = dist.get_world_size(group) if world is None else world
        self.rank = dist.get_rank(group) if rank is None else rank
        self.dst = dst
        self.params = params
        self.device = torch.device(device)
        self.slot_bytes = params.bin_size * params.bin_size * 4
        self.in_place = in_place
        self.frame = None
        self.inbox = None
        self.d_map = None
        self.set_scene(aabbs)
        if self.rank == dst:
            self.frame = torch.zeros(params.height * params.width * 4, dtype=torch.uint8, device=self.device)

    def set_scene(self, aabbs):
        """(Re)derive the tile list and its split over the ranks from the scene (same call on every rank)."""
        from . import scene_tiles
        import numpy as np
        p = self.params
        self.tiles = scene_tiles(p, aabbs)
        by = self.tiles >> 16
        self.blocks = [row_block(r, self.world, p.height, p.bin_size) for r in range(self.world)]
        self.first = [int(np.searchsorted(by, b // p.bin_size, side="left")) for b, _ in self.blocks]
        self.end = [int(np.searchsorted(by, -(-e // p.bin_size), side="left")) if e > b else self.first[r]
                    for r, (b, e) in enumerate(self.blocks)]
        self.counts = [e - f for f, e in zip(self.first, self.end)]
        self.d_tiles = torch.from_numpy(self.tiles.copy()).to(self.device)
        if self.rank == self.dst:
            # every rank's run, in list order, in one buffer (the root's own run is copied in locally)
            self.inbox = torch.zeros(max(len(self.tiles), 1) * self.slot_bytes, dtype=torch.uint8, device=self.device)
            if self.in_place:
                from . import scene_tile_map
                own = scene_tile_map(p, self.tiles)
                f, e = self.first[self.rank], self.end[self.rank]
                own[(own >= f) & (own < e)] = -1  # (the root's own tiles are already in place)
                self.tile_map = own
                self.d_map = torch.from_numpy(own).to(self.device)

    @property
    def max_rows(self):
        return max(e - b for b, e in self.blocks)

    def root_block(self):
        """in_place, root: its own rows of the assembled frame -- what it renders its block into."""
        b, e = self.blocks[self.rank]
        w4 = self.params.width * 4
        return self.frame[b * w4:e * w4]

    def bytes_sent(self, rank=None):
        r = self.rank if rank is None else rank
        return 0 if r == self.dst else self.counts[r] * self.slot_bytes

    def packed_buffer(self):
        """A send buffer for this rank's run of tiles."""
        return torch.zeros(max(self.counts[self.rank], 1) * self.slot_bytes, dtype=torch.uint8, device=self.device)

    def block_buffer(self):
        """A buffer for this rank's row block (the largest block's size, as FrameGather's)."""
        return torch.zeros(self.max_rows * self.params.width * 4, dtype=torch.uint8, device=self.device)

    # ---- the three steps of a frame --------------------------------------------------------------------------
    def pack(self, block, packed, stream=0):
        """This rank's run of tiles out of its rendered block (asynchronous on `stream` for device tensors)."""
        n, first = self.counts[self.rank], self.first[self.rank]
        if n == 0 or (self.in_place and self.rank == self.dst):
            return
        rows = self.blocks[self.rank]
        if block.is_cuda:
            from . import tiles_pack
            tiles_pack(self.params, self.d_tiles.data_ptr() + 4 * first, n, block.data_ptr(), rows, packed.data_ptr(), stream)
        else:
            _tiles_copy_cpu(self.params, self.tiles[first:first + n], block, rows, packed, pack=True)

    def exchange(self, packed, async_op=False):
        """Runs to `dst`. Returns a work handle (wait() before assemble()). On the root the own run is copied into
        its place in the inbox on the current stream."""
        sb = self.slot_bytes
        if packed.is_cuda and dist.get_backend(self.group) == "gloo":
            return self._exchange_via_host(packed)
        ops = []
        if self.rank == self.dst:
            f, n = self.first[self.rank], self.counts[self.rank]
            if n and not self.in_place:
                self.inbox[f * sb:(f + n) * sb].copy_(packed[:n * sb])
            for r in range(self.world):
                if r != self.dst and self.counts[r]:
                    ops.append(dist.P2POp(dist.irecv, self.inbox[self.first[r] * sb:self.end[r] * sb], r, self.group))
        elif self.counts[self.rank]:
            ops.append(dist.P2POp(dist.isend, packed[:self.counts[self.rank] * sb], self.dst, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        handle = _Works(works)
        if not async_op:
            handle.wait()
        return handle

    def _exchange_via_host(self, packed):
        """Test path (gloo moves host memory only): the same exchange staged through the host, synchronously."""
        sb = self.slot_bytes
        torch.cuda.current_stream().synchronize()
        if self.rank == self.dst:
            f, n = self.first[self.rank], self.counts[self.rank]
            if n and not self.in_place:
                self.inbox[f * sb:(f + n) * sb].copy_(packed[:n * sb])
            for r in range(self.world):
                if r != self.dst and self.counts[r]:
                    host = torch.empty(self.counts[r] * sb, dtype=torch.uint8)
                    dist.recv(host, src=r, group=self.group)
                    self.inbox[self.first[r] * sb:self.end[r] * sb].copy_(host)
        elif self.counts[self.rank]:
            dist.send(packed[:self.counts[self.rank] * sb].cpu(), dst=self.dst, group=self.group)
        return _Done()

    def assemble(self, stream=0):
        """Root: the background for the whole frame, then every tile into its place (asynchronous on `stream` for
        device tensors). The assembled frame is `self.frame`."""
        if self.rank != self.dst:
            return
        p = self.params
        if self.in_place:
            b, e = self.blocks[self.rank]
            others = [(0, b), (e, p.height)]
            if self.frame.is_cuda:
                from . import tiles_assemble
                for rows in others:
                    if rows[1] > rows[0]:
                        tiles_assemble(p, self.d_map.data_ptr(), self.inbox.data_ptr(), self.frame.data_ptr(), rows, stream)
            else:
                ch = int(float(p.background) * float(p.ambient))
                w4 = p.width * 4
                for lo, hi in others:
                    f = self.frame[lo * w4:hi * w4].view(-1, 4)
                    f[:, 0:3] = ch
                    f[:, 3] = 0
                for r in range(self.world):
                    if r != self.rank and self.counts[r]:
                        _tiles_copy_cpu(p, self.tiles[self.first[r]:self.end[r]], self.frame, (0, p.height),
                                        self.inbox[self.first[r] * self.slot_bytes:], pack=False)
            return
        if self.frame.is_cuda:
            from . import background_fill, tiles_unpack
            background_fill(p, self.frame.data_ptr(), p.height, stream)
            if len(self.tiles):
                tiles_unpack(p, self.d_tiles.data_ptr(), len(self.tiles), self.inbox.data_ptr(), self.frame.data_ptr(), stream)
        else:
            ch = int(float(p.background) * float(p.ambient))  # Color{background} * ambient, spr:8-16 (truncating)
            f = self.frame.view(-1, 4)
            f[:, 0:3] = ch
            f[:, 3] = 0
            _tiles_copy_cpu(p, self.tiles, self.frame, (0, p.height), self.inbox, pack=False)


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        return True


def _tiles_copy_cpu(params, tiles, frame_block, rows, packed, pack):
    """par_tiles_pack / par_tiles_unpack on host tensors (the CPU tests of the N > 1 path): frame_block holds rows
    [rows[0], rows[1]) of the frame, packed the slots."""
    B, W, H = params.bin_size, params.width, params.height
    fb = frame_block.view(torch.int32)[:(rows[1] - rows[0]) * W].view(rows[1] - rows[0], W)
    slots = packed.view(torch.int32)[:len(tiles) * B * B].view(len(tiles), B, B)
    for i, t in enumerate(tiles):
        bx, by = int(t) & 0xFFFF, int(t) >> 16
        c0, r0 = bx * B, by * B
        tw = min(B, W - c0)
        lo, hi = max(r0, rows[0]), min(r0 + B, H, rows[1])
        if tw <= 0 or hi <= lo:
            continue
        if pack:
            slots[i, lo - r0:hi - r0, :tw] = fb[lo - rows[0]:hi - rows[0], c0:c0 + tw]
        else:
            fb[lo - rows[0]:hi - rows[0], c0:c0 + tw] = slots[i, lo - r0:hi - r0, :tw]
