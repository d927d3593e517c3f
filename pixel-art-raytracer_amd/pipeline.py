"""Frames in flight.

One frame of the hot path is a chain of short, latency-bound kernels (hash build -> per-column records -> pixels):
on its own it leaves most of the chip idle. A renderer that produces a stream of frames therefore keeps several
frames in flight, each with its own context (spatial hash, work lists), HIP stream and output buffers, exactly as a
swap chain does; the kernels of consecutive frames then overlap on the device. Every frame is still one complete
pass of the reference's render call (alt:690-760) over the scene as it was when the frame was submitted.

This is host-side plumbing over the C ABI (one `par_context` per slot); a C++ host does the same with K contexts.
"""
import torch

from . import Renderer, plane_bytes


class FrameSlot:
    def __init__(self, renderer, stream, buffers, rows):
        self.renderer = renderer
        self.stream = stream
        self.buffers = buffers          # plane name -> torch.uint8 tensor
        self.ptrs = {k: v.data_ptr() for k, v in buffers.items()}
        self.rows = rows
        self.pending = None             # e.g. the collective that last read this slot's buffers


class FramePipeline:
    def __init__(self, params, aabbs, sprites, light, depth=4, device=0, rows=None, planes=("fb", "palidx"),
                 rows_alloc=None, sprite_ids=None):
        dev = torch.device("cuda", device)
        r0, r1 = rows or (0, params.height)
        n_rows = rows_alloc or (r1 - r0)
        self.params = params
        self.slots = []
        for _ in range(depth):
            r = Renderer(params, device)
            r.set_scene(aabbs, sprites, light, sprite_ids)
            bufs = {k: torch.zeros(n_rows * params.width * plane_bytes(k), dtype=torch.uint8, device=dev)
                    for k in planes}
            self.slots.append(FrameSlot(r, torch.cuda.Stream(device=dev), bufs, (r0, r1)))

    def slot(self, i):
        return self.slots[i % len(self.slots)]

    def submit(self, i, flags=0):
        """Enqueue frame i on its slot's stream (asynchronous). Returns the slot."""
        s = self.slot(i)
        s.renderer.render_device(s.ptrs, rows=s.rows, flags=flags, stream=s.stream.cuda_stream)
        return s

    def update_aabbs(self, i, aabbs, first=0, light=None):
        """Scene mutation for frame i (alt:643-678): applied to the slot that will render it."""
        s = self.slot(i)
        s.stream.synchronize()
        s.renderer.update_aabbs(aabbs, first)
        if light is not None:
            s.renderer.set_light(light)

    def synchronize(self):
        for s in self.slots:
            s.stream.synchronize()

    def close(self):
        self.synchronize()
        for s in self.slots:
            s.renderer.close()
