"""The `pipeline` command over a batch of frames (reference src/cli/pipeline.zig:153-179: every input image goes through the recipe's
steps in order), resident on the device: `Pipeline(steps).run(frames)` with frames a torch tensor (n, rows, cols[, channels]).

One call of the C ABI (zg_batch_pipeline) per batch: each step is a single launch over all frames where the library has a batched
kernel, fused with its neighbour where it has a fused one, and equals the per-frame `Image` methods bit for bit."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .image import _PIXEL_BY_LAYOUT, Interpolation

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_LAYOUT_BY_PIXEL = {v: k for k, v in _PIXEL_BY_LAYOUT.items()}


class Step:
    """One recipe step; the constructors mirror the CLI's step options (resize: src/cli/resize.zig; blur with its six types:
    src/cli/blur.zig:98-170; edges with its three detectors: src/cli/edges.zig:85-135) plus convert and warp."""

    def __init__(self, c_step: L.ZgStep, keep=None):
        self.c = c_step
        self._keep = keep  # host arrays the C struct points at

    @staticmethod
    def gaussian_blur(sigma: float) -> "Step":
        s = L.ZgStep()
        s.kind, s.sigma = L.STEP_GAUSSIAN_BLUR, float(sigma)
        return Step(s)

    @staticmethod
    def box_blur(radius: int) -> "Step":
        s = L.ZgStep()
        s.kind, s.radius = L.STEP_BOX_BLUR, int(radius)
        return Step(s)

    @staticmethod
    def median_blur(radius: int = 1) -> "Step":
        """blur --type median (blur.zig:116-123): Image.medianBlur(radius)."""
        s = L.ZgStep()
        s.kind, s.radius = L.STEP_MEDIAN_BLUR, int(radius)
        return Step(s)

    @staticmethod
    def motion_blur_linear(angle: float = 0.0, distance: int = 10, cos_sin=None) -> "Step":
        """blur --type motion_linear (blur.zig:124-146): Image.motionBlur(.{ .linear = .{ .angle (radians), .distance } }). cos_sin: the host's
        own (cos, sin) of the angle, as in Image.motion_blur_linear."""
        from .image import _cos_sin
        ca, sa = cos_sin if cos_sin is not None else _cos_sin(angle)
        s = L.ZgStep()
        s.kind, s.motion, s.angle, s.cos_a, s.sin_a, s.distance = L.STEP_MOTION_BLUR, L.MOTION_LINEAR, float(angle), float(ca), float(sa), int(distance)
        return Step(s)

    @staticmethod
    def motion_blur_radial(center_x: float = 0.5, center_y: float = 0.5, strength: float = 0.5, spin: bool = False) -> "Step":
        """blur --type motion_zoom / motion_spin (blur.zig:147-170)."""
        s = L.ZgStep()
        s.kind, s.motion = L.STEP_MOTION_BLUR, (L.MOTION_RADIAL_SPIN if spin else L.MOTION_RADIAL_ZOOM)
        s.center_x, s.center_y, s.strength = float(center_x), float(center_y), float(strength)
        return Step(s)

    @staticmethod
    def edges_sobel() -> "Step":
        """edges --filter sobel through the pipeline's grey bridge (edges.zig:126-135): the frames keep their type."""
        s = L.ZgStep()
        s.kind, s.edges = L.STEP_EDGES, L.EDGES_SOBEL
        return Step(s)

    @staticmethod
    def edges_canny(sigma: float = 1.0, low: float = 50.0, high: float = 100.0) -> "Step":
        s = L.ZgStep()
        s.kind, s.edges, s.sigma, s.low, s.high = L.STEP_EDGES, L.EDGES_CANNY, float(sigma), float(low), float(high)
        return Step(s)

    @staticmethod
    def edges_shen_castan(smooth: float = 0.9, window_size: int = 7, high_ratio: float = 0.99, low_rel: float = 0.5, use_nms: bool = False) -> "Step":
        s = L.ZgStep()
        s.kind, s.edges, s.sigma, s.window, s.high, s.low, s.use_nms = L.STEP_EDGES, L.EDGES_SHEN_CASTAN, float(smooth), int(window_size), float(high_ratio), float(low_rel), int(bool(use_nms))
        return Step(s)

    @staticmethod
    def resize(rows: int, cols: int, method: Interpolation = Interpolation.bilinear) -> "Step":
        s = L.ZgStep()
        s.kind, s.out_rows, s.out_cols, s.method = L.STEP_RESIZE, int(rows), int(cols), method._c()
        return Step(s)

    @staticmethod
    def convert(dst_space: int, dtype=np.float32, srgb_lut=None) -> "Step":
        ch = 1 if dst_space == L.CS_GRAY else (4 if dst_space == L.CS_RGBA else 3)
        s = L.ZgStep()
        s.kind, s.dst_space, s.dst_pixel = L.STEP_CONVERT, int(dst_space), _PIXEL_BY_LAYOUT[(np.dtype(dtype).name, ch)]
        keep = None
        if srgb_lut is not None:
            keep = np.ascontiguousarray(srgb_lut, np.float32)
            s.srgb_lut = keep.ctypes.data
        return Step(s, keep)

    @staticmethod
    def warp(transform, rows: int, cols: int, method: Interpolation = Interpolation.bilinear) -> "Step":
        s = L.ZgStep()
        s.kind, s.out_rows, s.out_cols, s.method, s.transform = L.STEP_WARP, int(rows), int(cols), method._c(), int(transform.kind)
        for i, v in enumerate(transform.coefficients()):
            s.m[i] = float(v)
        return Step(s)


class Pipeline:
    def __init__(self, steps: Sequence[Step]):
        self.steps: List[Step] = list(steps)
        self._c = (L.ZgStep * max(1, len(self.steps)))(*[st.c for st in self.steps])

    def out_layout(self, rows: int, cols: int, pixel: int, space: int):
        r, c, p, sp = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int()
        L.check(L.lib().zg_batch_pipeline_shape(rows, cols, pixel, space, self._c, len(self.steps), C.byref(r), C.byref(c), C.byref(p), C.byref(sp)))
        return r.value, c.value, p.value, sp.value

    def run(self, frames, space: Optional[int] = None, out=None):
        """frames: CUDA tensor (n, rows, cols) or (n, rows, cols, channels), contiguous. Returns the (n, rows', cols'[, channels']) result."""
        if torch is None or not isinstance(frames, torch.Tensor) or not frames.is_cuda:
            raise ValueError("Pipeline.run takes device frames (a CUDA torch tensor); the host-pointer layer is per image")
        if frames.ndim not in (3, 4) or not frames.is_contiguous():
            raise ValueError("expected contiguous frames (n, rows, cols[, channels])")
        n, rows, cols = (int(v) for v in frames.shape[:3])
        ch = 1 if frames.ndim == 3 else int(frames.shape[3])
        pixel = _PIXEL_BY_LAYOUT[(str(frames.dtype).replace("torch.", ""), ch)]
        if space is None:
            space = {1: L.CS_GRAY, 3: L.CS_RGB, 4: L.CS_RGBA}[ch]
        orows, ocols, opixel, _ = self.out_layout(rows, cols, pixel, space)
        odtype, och = _LAYOUT_BY_PIXEL[opixel]
        shape = (n, orows, ocols) if och == 1 else (n, orows, ocols, och)
        tdtype = {"uint8": torch.uint8, "float32": torch.float32}[odtype]
        if out is None:
            out = torch.empty(shape, dtype=tdtype, device=frames.device)
        elif tuple(out.shape) != shape or out.dtype != tdtype or not out.is_contiguous() or out.device != frames.device:
            raise L.DimensionMismatch(f"out must be a contiguous {tdtype} tensor of shape {shape} on {frames.device}")
        with torch.cuda.device(frames.device):
            L.check(L.lib().zg_batch_pipeline(C.c_void_p(frames.data_ptr()), n, rows, cols, pixel, int(space), self._c, len(self.steps),
                                              C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(frames.device).cuda_stream)))
        return out

    def run_multi(self, ctx: "Multi", frames, space: Optional[int] = None, out=None):
        """`run` over every GPU of a zg_multi context (zg_multi_batch_pipeline): `frames` and the result live on the context's root device,
        frames shard in contiguous blocks with no halo and no collective on the data path; returns when the results are complete.
        Returns (result, (scatter_ms, busiest_device_kernels_ms, whole_call_ms))."""
        if torch is None or not isinstance(frames, torch.Tensor) or not frames.is_cuda or frames.ndim not in (3, 4) or not frames.is_contiguous():
            raise ValueError("expected contiguous device frames (n, rows, cols[, channels])")
        if frames.device.index != ctx.root_device:  # the root's kernels and RCCL would take the pointer for one of their own device's
            raise ValueError(f"frames live on {frames.device}, the context's root device is cuda:{ctx.root_device}")
        n, rows, cols = (int(v) for v in frames.shape[:3])
        ch = 1 if frames.ndim == 3 else int(frames.shape[3])
        pixel = _PIXEL_BY_LAYOUT[(str(frames.dtype).replace("torch.", ""), ch)]
        if space is None:
            space = {1: L.CS_GRAY, 3: L.CS_RGB, 4: L.CS_RGBA}[ch]
        orows, ocols, opixel, _ = self.out_layout(rows, cols, pixel, space)
        odtype, och = _LAYOUT_BY_PIXEL[opixel]
        shape = (n, orows, ocols) if och == 1 else (n, orows, ocols, och)
        tdtype = {"uint8": torch.uint8, "float32": torch.float32}[odtype]
        if out is None:
            out = torch.empty(shape, dtype=tdtype, device=frames.device)
        elif tuple(out.shape) != shape or out.dtype != tdtype or not out.is_contiguous() or out.device != frames.device:
            raise L.DimensionMismatch(f"out must be a contiguous {tdtype} tensor of shape {shape} on {frames.device}")
        times = (C.c_float * 3)()
        ctx.wait_stream(torch.cuda.current_stream(frames.device).cuda_stream)  # what filled `frames` was enqueued on torch's current stream
        L.check(L.lib().zg_multi_batch_pipeline(ctx.handle, C.c_void_p(frames.data_ptr()), n, rows, cols, pixel, int(space), self._c, len(self.steps),
                                                C.c_void_p(out.data_ptr()), times))
        return out, tuple(times)


class Multi:
    """A zg_multi context: ONE host thread drives several GPUs (include/zignal_hip.h; zg_multi.cpp). devices=None means every visible device."""

    def __init__(self, devices: Optional[Sequence[int]] = None):
        h = C.c_void_p()
        self.handle = None
        if devices is None:
            L.check(L.lib().zg_multi_create(None, 0, C.byref(h)))
        else:
            arr = (C.c_int * len(devices))(*devices)
            L.check(L.lib().zg_multi_create(arr, len(devices), C.byref(h)))
        self.handle = h
        self.root_device = int(devices[0]) if devices else 0  # dev[0] of the context: where the caller's frames and results live

    def __del__(self):  # a context that was never closed still gives back its streams, communicators and staging buffers
        try:
            self.close()
        except Exception:
            pass

    def device_count(self) -> int:
        return int(L.lib().zg_multi_device_count(self.handle))

    def wait_stream(self, stream: int) -> None:
        L.check(L.lib().zg_multi_wait_stream(self.handle, C.c_void_p(stream)))

    def close(self) -> None:
        if self.handle:
            L.lib().zg_multi_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def piece_range(n_frames: int, world: int, chunks: int, device: int, piece: int) -> Tuple[int, int]:
    """zg_multi's piece arithmetic (host only): frames [begin, end) of the batch that form piece `piece` of device `device`'s shard."""
    b, e = C.c_uint32(), C.c_uint32()
    L.check(L.lib().zg_multi_piece_range(n_frames, world, chunks, device, piece, C.byref(b), C.byref(e)))
    return b.value, e.value
