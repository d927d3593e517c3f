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
