"""Frames in flight.

One frame of the hot path is a chain of short, latency-bound kernels (hash build -> per-column records -> pixels):
on its own it leaves most of the chip idle. A renderer that produces a stream of frames therefore keeps several
frames in flight, each with its own context (spatial hash, work lists), HIP stream and output buffers, exactly as a
swap chain does; the kernels of consecutive frames then overlap on the device. Every frame is still one complete
pass of the reference's render call (alt:690-760) over the scene as it was when the frame was submitted.

This is host-side plumbing over the C ABI (one `par_context` per slot); a C++ host does the same with K contexts.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import RENDER_PIPELINED, Renderer, lib, plane_bytes
from .types import Outputs


class FrameSlot:
    def __init__(self, renderer, stream, buffers, rows):
        self.renderer = renderer
        self.stream = stream
        self.buffers = buffers          # plane name -> torch.uint8 tensor
        self.ptrs = {k: v.data_ptr() for k, v in buffers.items()}
        self.rows = rows
        self.pending = None             # e.g. the collective that last read this slot's buffers


class FramePipeline:
    """`depth` frames in flight. HIP streams share a few hardware queues (4 by default on ROCm), and two streams on
    the same queue do not overlap at all; which stream lands on which queue depends on what the process created
    before. So the streams are chosen by measurement: from a pool of candidates, keep those that overlap with every
    stream kept so far (a few dozen small probe frames each)."""

    def __init__(self, params, aabbs, sprites, light, depth=4, device=0, rows=None, planes=("fb", "palidx"),
                 rows_alloc=None, sprite_ids=None, calibrate=True, fb_targets=None):
        """`fb_targets`: one caller-owned uint8 tensor per slot to render the RGBA block into (e.g. the slot's rows of
        an assembled frame) instead of a buffer of the pipeline's own."""
        dev = torch.device("cuda", device)
        r0, r1 = rows or (0, params.height)
        n_rows = rows_alloc or (r1 - r0)
        self.params = params
        self.slots = []
        for k_slot in range(depth):
            r = Renderer(params, device)
            r.set_scene(aabbs, sprites, light, sprite_ids)
            bufs = {k: torch.zeros(n_rows * params.width * plane_bytes(k), dtype=torch.uint8, device=dev)
                    for k in planes if not (k == "fb" and fb_targets)}
            if fb_targets and "fb" in planes:
                if fb_targets[k_slot].numel() < (r1 - r0) * params.width * plane_bytes("fb"):
                    raise ValueError("fb target smaller than the row block")
                bufs["fb"] = fb_targets[k_slot]
            self.slots.append(FrameSlot(r, None, bufs, (r0, r1)))
        candidates = [torch.cuda.Stream(device=dev) for _ in range(depth if depth < 2 or not calibrate else 3 * depth)]
        chosen = self._pick_streams(candidates, depth) if calibrate and depth > 1 else candidates[:depth]
        for s, stream in zip(self.slots, chosen):
            s.stream = stream
        self.streams_overlap = len(chosen) == depth and getattr(self, "_all_overlap", True)

    def _probe(self, sa, sb, n=24):
        """Wall time of n frames alternating between slots 0 and 1 on streams sa and sb."""
        a, b = self.slots[0], self.slots[1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            s, st = (a, sa) if i % 2 == 0 else (b, sb)
            s.renderer.render_device(s.ptrs, rows=s.rows, stream=st.cuda_stream)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def _pick_streams(self, candidates, depth):
        first = candidates[0]
        self._probe(first, first)  # warm-up
        serial = min(self._probe(first, first) for _ in range(2))
        chosen = [first]
        for c in candidates[1:]:
            if len(chosen) == depth:
                break
            if all(min(self._probe(x, c), self._probe(x, c)) < 0.85 * serial for x in chosen):
                chosen.append(c)
        self._all_overlap = len(chosen) == depth
        for c in candidates:  # not enough distinct queues (or frames too short to tell): take what is there
            if len(chosen) == depth:
                break
            if c not in chosen:
                chosen.append(c)
        return chosen

    def slot(self, i):
        return self.slots[i % len(self.slots)]

    def submit_many(self, first, n, flags=0, threads=False):
        """Enqueue frames first .. first + n - 1 round-robin over the slots with ONE call into the library
        (par_render_device_slots): the render loop of the swap chain runs in C, not in the interpreter.
        `threads`: one submitting thread per slot instead (each calls the library for its own slot's frames; a slot's
        context is only ever touched by its own thread, and ctypes releases the interpreter lock for the call): small
        frames are bound by the host's launches -- four launches of 3-4 us per graybox frame against 10 us of device
        time -- and how fast one host thread is differs from box to box (host/par_pipeline.cpp --threads does the same)."""
        if threads and len(self.slots) > 1 and n >= 4 * len(self.slots):
            return self._submit_threaded(first, n, flags)
        if getattr(self, "_slot_args", None) is None:
            k = len(self.slots)
            ctxs = (C.c_void_p * k)(*[s.renderer._ctx for s in self.slots])
            streams = (C.c_void_p * k)(*[s.stream.cuda_stream for s in self.slots])
            outs = (Outputs * k)(*[Outputs(*[s.ptrs.get(p) for p in ("fb", "gbuf", "palidx", "brightness", "lit")])
                                   for s in self.slots])
            self._slot_args = (ctxs, streams, outs)
        ctxs, streams, outs = self._slot_args
        r0, r1 = self.slots[0].rows
        rc = lib().par_render_device_slots(ctxs, streams, outs, len(self.slots), r0, r1, first, n, flags)
        if rc != 0:
            raise RuntimeError(f"par_render_device_slots failed with status {rc}")

    def _submit_threaded(self, first, n, flags):
        from concurrent.futures import ThreadPoolExecutor
        k = len(self.slots)
        if getattr(self, "_pool", None) is None:
            self._pool = ThreadPoolExecutor(k)
            self._one = []
            for s in self.slots:
                self._one.append(((C.c_void_p * 1)(s.renderer._ctx), (C.c_void_p * 1)(s.stream.cuda_stream),
                                  (Outputs * 1)(Outputs(*[s.ptrs.get(p) for p in ("fb", "gbuf", "palidx", "brightness", "lit")]))))
        r0, r1 = self.slots[0].rows
        device = self.slots[0].stream.device

        def run(j, count):
            torch.cuda.set_device(device)
            ctxs, streams, outs = self._one[j]
            return lib().par_render_device_slots(ctxs, streams, outs, 1, r0, r1, 0, count, flags | RENDER_PIPELINED)

        counts = [len(range(first + ((j - first) % k), first + n, k)) for j in range(k)]  # frames f with f % k == j
        for rc in self._pool.map(run, range(k), counts):
            if rc != 0:
                raise RuntimeError(f"par_render_device_slots failed with status {rc}")

    def kernel_spans_us(self):
        """Per kernel of the frame, the mean span (first workgroup start to last workgroup end, microseconds) over the
        slots' most recent frames rendered with flag bit 29 (needs PAR_DEBUG_STAMPS=1 when the contexts were made)."""
        names = ("build_fill", "resolve_fill", "columns_fill", "render_items", "render_overflow", "render_tiles")
        rows, wgs = 6, 8192
        spans = {k: [] for k in names}
        buf = np.zeros(rows * wgs * 8, dtype=np.uint64)
        for s in self.slots:
            rc = lib().par_debug_read_stamps(s.renderer._ctx, buf.ctypes.data_as(C.c_void_p), buf.size)
            if rc != 0:
                return None
            st = buf.reshape(rows, wgs, 8)
            for r, k in enumerate(names):
                b, e = st[r, :, 0], st[r, :, 7]
                live = b > 0
                if live.any():
                    spans[k].append((max(int(e[live].max()), int(b[live].max())) - int(b[live].min())) * 0.01)
        return {k: round(float(np.mean(v)), 3) for k, v in spans.items() if v}

    def submit(self, i, flags=0):
        """Enqueue frame i on its slot's stream (asynchronous). Returns the slot."""
        s = self.slot(i)
        if len(self.slots) > 1:
            flags |= RENDER_PIPELINED  # several frames in flight: throughput before latency
        s.renderer.render_device(s.ptrs, rows=s.rows, flags=flags, stream=s.stream.cuda_stream)
        return s

    def update_aabbs(self, i, aabbs, first=0, light=None):
        """Scene mutation for frame i (alt:643-678): applied to the slot that will render it."""
        s = self.slot(i)
        s.renderer.update_aabbs(aabbs, first, stream=s.stream.cuda_stream)  # in stream order, no wait
        if light is not None:
            s.renderer.set_light(light)  # (host state: it travels with the next frame's kernel arguments)

    def synchronize(self):
        for s in self.slots:
            s.stream.synchronize()

    def close(self):
        self.synchronize()
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown()
            self._pool = None
        for s in self.slots:
            s.renderer.close()
