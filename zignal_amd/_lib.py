"""ctypes binding of libzignal_hip.so — the C ABI declared in include/zignal_hip.h.

There is no CPU fallback anywhere in this package: if the HIP library is missing or an entry point
fails, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZIGNAL_HIP_LIBRARY") or os.path.join(_HERE, "libzignal_hip.so")  # the override is for A/B runs of two builds

# enums (ordinals as in include/zignal_hip.h == the reference's declaration order)
PIXEL_U8, PIXEL_F32, PIXEL_RGB_U8, PIXEL_RGBA_U8, PIXEL_RGB_F32, PIXEL_RGBA_F32 = range(6)
BORDER_ZERO, BORDER_REPLICATE, BORDER_MIRROR, BORDER_WRAP = range(4)
INTERP_NEAREST, INTERP_BILINEAR, INTERP_BICUBIC, INTERP_CATMULL_ROM, INTERP_MITCHELL, INTERP_LANCZOS = range(6)
TRANSFORM_SIMILARITY, TRANSFORM_AFFINE, TRANSFORM_PROJECTIVE = range(3)
CS_GRAY, CS_RGB, CS_RGBA, CS_OKLAB, CS_XYZ, CS_YCBCR, CS_HSL, CS_HSV, CS_LAB, CS_LCH, CS_LMS, CS_OKLCH, CS_XYB = range(13)

OK, ERR_DIMENSION_MISMATCH, ERR_INVALID_ARGUMENT, ERR_OUT_OF_MEMORY, ERR_HIP, ERR_UNSUPPORTED, ERR_CODEC = range(7)


class ZgImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_size_t), ("rows", C.c_uint32),
                ("cols", C.c_uint32), ("pixel", C.c_int32)]


class ZgMethod(C.Structure):
    _fields_ = [("kind", C.c_int32), ("b", C.c_float), ("c", C.c_float), ("lanczos_lut", C.c_void_p)]


class ZgStep(C.Structure):
    """zg_step: one step of zg_batch_pipeline (the CLI's `pipeline` recipe steps, src/cli/pipeline.zig:20-24, plus convert and warp)."""
    _fields_ = [("kind", C.c_int), ("sigma", C.c_float), ("radius", C.c_uint32), ("out_rows", C.c_uint32), ("out_cols", C.c_uint32),
                ("method", ZgMethod), ("dst_pixel", C.c_int), ("dst_space", C.c_int), ("srgb_lut", C.c_void_p), ("transform", C.c_int),
                ("m", C.c_float * 9), ("motion", C.c_int), ("angle", C.c_float), ("cos_a", C.c_float), ("sin_a", C.c_float), ("distance", C.c_uint32),
                ("center_x", C.c_float), ("center_y", C.c_float), ("strength", C.c_float), ("edges", C.c_int), ("low", C.c_float), ("high", C.c_float),
                ("window", C.c_uint32), ("use_nms", C.c_int)]


STEP_GAUSSIAN_BLUR, STEP_BOX_BLUR, STEP_RESIZE, STEP_CONVERT, STEP_WARP, STEP_MEDIAN_BLUR, STEP_MOTION_BLUR, STEP_EDGES = range(8)
MOTION_LINEAR, MOTION_RADIAL_ZOOM, MOTION_RADIAL_SPIN = range(3)
EDGES_SOBEL, EDGES_CANNY, EDGES_SHEN_CASTAN = range(3)


class ZgPngHeader(C.Structure):
    """zg_png_header == png.Header (png.zig:135-149)."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bit_depth", C.c_uint8), ("color_type", C.c_uint8),
                ("compression_method", C.c_uint8), ("filter_method", C.c_uint8), ("interlace_method", C.c_uint8),
                ("has_gamma", C.c_uint8), ("has_srgb", C.c_uint8), ("srgb_intent", C.c_uint8), ("gamma", C.c_float)]


class ZgPngLimits(C.Structure):
    """zg_png_limits == png.DecodeLimits (png.zig:23-41)."""
    _fields_ = [("max_png_bytes", C.c_size_t), ("max_chunk_bytes", C.c_size_t), ("max_idat_bytes", C.c_size_t),
                ("max_chunks", C.c_size_t), ("max_width", C.c_uint32), ("max_height", C.c_uint32),
                ("max_pixels", C.c_uint64), ("max_decompressed_bytes", C.c_size_t)]


class ZgPngEncodeOptions(C.Structure):
    """zg_png_encode_options == png.EncodeOptions (png.zig:1296-1316)."""
    _fields_ = [("filter", C.c_int), ("compression_level", C.c_int), ("has_gamma", C.c_int), ("gamma", C.c_float),
                ("srgb_intent", C.c_int)]


class ZgJpegHeader(C.Structure):
    """zg_jpeg_header == jpeg.Header (jpeg.zig:61-74)."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("precision", C.c_uint8), ("num_components", C.c_uint8),
                ("progressive", C.c_uint8), ("subsampling", C.c_int8)]


class ZgJpegLimits(C.Structure):
    """zg_jpeg_limits == jpeg.DecodeLimits (jpeg.zig:19-33)."""
    _fields_ = [("max_jpeg_bytes", C.c_size_t), ("max_marker_bytes", C.c_size_t), ("max_width", C.c_uint32), ("max_height", C.c_uint32),
                ("max_pixels", C.c_uint64), ("max_blocks", C.c_size_t), ("max_scans", C.c_size_t)]


class ZgJpegEncodeOptions(C.Structure):
    """zg_jpeg_encode_options == jpeg.EncodeOptions (jpeg.zig:284-290)."""
    _fields_ = [("quality", C.c_int), ("subsampling", C.c_int), ("density_dpi", C.c_int), ("comment", C.c_char_p), ("comment_len", C.c_size_t)]


class ZignalError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"zignal_hip status {status}: {message}")
        self.status = status


class DimensionMismatch(ZignalError):
    """error.DimensionMismatch (reference src/image.zig:636,927,947,962)."""


class InvalidArgument(ZignalError, ValueError):
    """error.InvalidSigma / InvalidScaleFactor / InvalidDimensions."""


class CodecError(ZignalError):
    """An error of a codec's error set (src/codecs/png.zig); `.name` is the Zig error name, e.g. "InvalidCrc"."""

    def __init__(self, status: int, message: str):
        super().__init__(status, message)
        self.name = message.split(" ", 1)[0]


_lib = None

_IMG = C.POINTER(ZgImage)
_METHOD = C.POINTER(ZgMethod)
_F32P = C.POINTER(C.c_float)
_U32P = C.POINTER(C.c_uint32)

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "zg_init": [C.c_int],
    "zg_shutdown": [],
    "zg_last_error": [],
    "zg_version": [],
    "zg_device_count": [],
    "zg_malloc": [C.POINTER(C.c_void_p), C.c_size_t],
    "zg_free": [C.c_void_p],
    "zg_memcpy_h2d": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    "zg_memcpy_d2h": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    "zg_set_device": [C.c_int],
    "zg_get_device": [C.POINTER(C.c_int)],
    "zg_malloc_host": [C.POINTER(C.c_void_p), C.c_size_t],
    "zg_free_host": [C.c_void_p],
    "zg_memcpy_h2d_async": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    "zg_memcpy_d2h_async": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    "zg_image_upload": [_IMG, _IMG, C.c_void_p],
    "zg_image_download": [_IMG, _IMG, C.c_void_p],
    "zg_stream_wait_event": [C.c_void_p, C.c_void_p],
    "zg_event_create": [C.POINTER(C.c_void_p)],
    "zg_event_destroy": [C.c_void_p],
    "zg_event_record": [C.c_void_p, C.c_void_p],
    "zg_event_synchronize": [C.c_void_p],
    "zg_event_elapsed_ms": [C.c_void_p, C.c_void_p, _F32P],
    "zg_graph_begin_capture": [C.c_void_p],
    "zg_graph_end_capture": [C.c_void_p, C.POINTER(C.c_void_p)],
    "zg_graph_launch": [C.c_void_p, C.c_void_p],
    "zg_graph_destroy": [C.c_void_p],
    "zg_release_graph_scratch": [],
    "zg_trim_scratch": [],
    "zg_devmath_apply": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    "zg_lanczos_plane_weights": [C.c_uint32, C.c_uint32, _F32P],
    "zg_resize_lanczos_weights": [_IMG, _IMG, _F32P, _F32P, C.c_void_p],
    "zg_resize_lanczos_weights_host": [_IMG, _IMG, _F32P, _F32P],
    "zg_resize_convert": [_IMG, C.c_int, _IMG, C.c_int, _METHOD, _F32P, C.c_void_p],
    "zg_resize_convert_host": [_IMG, C.c_int, _IMG, C.c_int, _METHOD, _F32P],
    "zg_batch_pipeline_shape": [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(ZgStep), C.c_uint32, _U32P, _U32P, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "zg_batch_pipeline": [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(ZgStep), C.c_uint32, C.c_void_p, C.c_void_p],
    "zg_pyramid_build": [_IMG, _IMG, _F32P, C.c_uint32, C.c_void_p],
    "zg_multi_create": [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)],
    "zg_multi_destroy": [C.c_void_p],
    "zg_multi_device_count": [C.c_void_p],
    "zg_multi_wait_stream": [C.c_void_p, C.c_void_p],
    "zg_multi_batch_blur_resize": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.c_void_p, C.c_uint32, C.c_uint32, _METHOD, _F32P],
    "zg_multi_batch_pipeline": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(ZgStep), C.c_uint32, C.c_void_p, _F32P],
    "zg_multi_piece_range": [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, _U32P, _U32P],
    "zg_sizeof_step": [],
    "zg_stream_create": [C.POINTER(C.c_void_p)],
    "zg_stream_destroy": [C.c_void_p],
    "zg_stream_synchronize": [C.c_void_p],
    "zg_pixel_size": [C.c_int],
    "zg_conv_separable": [_IMG, _IMG, _F32P, C.c_uint32, _F32P, C.c_uint32, C.c_int, C.c_void_p],
    "zg_conv_separable_host": [_IMG, _IMG, _F32P, C.c_uint32, _F32P, C.c_uint32, C.c_int],
    "zg_gaussian_blur": [_IMG, _IMG, C.c_float, C.c_void_p],
    "zg_gaussian_blur_host": [_IMG, _IMG, C.c_float],
    "zg_gaussian_kernel": [C.c_float, _F32P, C.c_uint32],
    "zg_conv_separable_planes": [_IMG, _IMG, C.c_uint32, _F32P, C.c_uint32, _F32P, C.c_uint32, C.c_int, C.c_void_p],
    "zg_gaussian_blur_planes": [_IMG, _IMG, C.c_uint32, C.c_float, C.c_void_p],
    "zg_convolve": [_IMG, _IMG, _F32P, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p],
    "zg_convolve_host": [_IMG, _IMG, _F32P, C.c_uint32, C.c_uint32, C.c_int],
    "zg_box_blur": [_IMG, _IMG, C.c_uint32, C.c_void_p],
    "zg_box_blur_host": [_IMG, _IMG, C.c_uint32],
    "zg_resize": [_IMG, _IMG, _METHOD, C.c_void_p],
    "zg_resize_host": [_IMG, _IMG, _METHOD],
    "zg_letterbox": [_IMG, _IMG, _METHOD, _U32P, C.c_void_p],
    "zg_letterbox_host": [_IMG, _IMG, _METHOD, _U32P],
    "zg_warp": [_IMG, _IMG, C.c_int, _F32P, _METHOD, C.c_void_p],
    "zg_warp_host": [_IMG, _IMG, C.c_int, _F32P, _METHOD],
    "zg_rotate_into": [_IMG, _IMG, C.c_float, C.c_float, C.c_float, _METHOD, C.c_int, C.c_void_p],
    "zg_rotate_into_host": [_IMG, _IMG, C.c_float, C.c_float, C.c_float, _METHOD, C.c_int],
    "zg_rotate_bounds": [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, _U32P, _U32P],
    "zg_extract": [_IMG, _IMG, _F32P, C.c_float, C.c_float, C.c_float, _METHOD, C.c_int, C.c_void_p],
    "zg_extract_host": [_IMG, _IMG, _F32P, C.c_float, C.c_float, C.c_float, _METHOD, C.c_int],
    "zg_crop": [_IMG, _IMG, _F32P, C.c_void_p],
    "zg_crop_host": [_IMG, _IMG, _F32P],
    "zg_crop_dims": [_F32P, _U32P, _U32P],
    "zg_flip_left_right": [_IMG, C.c_void_p],
    "zg_flip_top_bottom": [_IMG, C.c_void_p],
    "zg_flip_left_right_host": [_IMG],
    "zg_flip_top_bottom_host": [_IMG],
    "zg_insert": [_IMG, _IMG, _F32P, C.c_float, C.c_float, C.c_float, _METHOD, C.c_int, C.c_void_p],
    "zg_insert_host": [_IMG, _IMG, _F32P, C.c_float, C.c_float, C.c_float, _METHOD, C.c_int],
    "zg_copy": [_IMG, _IMG, C.c_void_p],
    "zg_fill": [_IMG, C.c_void_p, C.c_void_p],
    "zg_set_border": [_IMG, _U32P, C.c_void_p, C.c_void_p],
    "zg_fill_host": [_IMG, C.c_void_p],
    "zg_set_border_host": [_IMG, _U32P, C.c_void_p],
    "zg_convert": [_IMG, C.c_int, _IMG, C.c_int, _F32P, C.c_void_p],
    "zg_convert_host": [_IMG, C.c_int, _IMG, C.c_int, _F32P],
    "zg_sharpen": [_IMG, _IMG, C.c_uint32, C.c_void_p],
    "zg_sharpen_host": [_IMG, _IMG, C.c_uint32],
    "zg_integral": [_IMG, _F32P, C.c_void_p],
    "zg_integral_host": [_IMG, _F32P],
    "zg_invert": [_IMG, C.c_void_p],
    "zg_invert_host": [_IMG],
    "zg_order_statistic_blur": [_IMG, _IMG, C.c_uint32, C.c_int, C.c_double, C.c_int, C.c_void_p],
    "zg_order_statistic_blur_host": [_IMG, _IMG, C.c_uint32, C.c_int, C.c_double, C.c_int],
    "zg_autocontrast": [_IMG, C.c_float, C.c_void_p],
    "zg_autocontrast_host": [_IMG, C.c_float],
    "zg_equalize": [_IMG, C.c_void_p],
    "zg_equalize_host": [_IMG],
    "zg_threshold_otsu": [_IMG, _IMG, C.POINTER(C.c_uint8), C.c_void_p],
    "zg_threshold_otsu_host": [_IMG, _IMG, C.POINTER(C.c_uint8)],
    "zg_threshold_adaptive_mean": [_IMG, _IMG, C.c_uint32, C.c_float, C.c_void_p],
    "zg_threshold_adaptive_mean_host": [_IMG, _IMG, C.c_uint32, C.c_float],
    "zg_morph": [_IMG, _IMG, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p],
    "zg_morph_host": [_IMG, _IMG, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int],
    "zg_sobel": [_IMG, _IMG, C.c_void_p],
    "zg_sobel_host": [_IMG, _IMG],
    "zg_canny": [_IMG, _IMG, C.c_float, C.c_float, C.c_float, C.c_void_p],
    "zg_canny_host": [_IMG, _IMG, C.c_float, C.c_float, C.c_float],
    "zg_isef_smooth": [_IMG, _IMG, C.c_float, C.c_void_p],
    "zg_shen_castan": [_IMG, _IMG, C.c_float, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p],
    "zg_shen_castan_host": [_IMG, _IMG, C.c_float, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int],
    "zg_motion_blur_linear": [_IMG, _IMG, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_void_p],
    "zg_motion_blur_linear_host": [_IMG, _IMG, C.c_float, C.c_float, C.c_float, C.c_uint32],
    "zg_motion_blur_radial": [_IMG, _IMG, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p],
    "zg_motion_blur_radial_host": [_IMG, _IMG, C.c_float, C.c_float, C.c_float, C.c_int],
    "zg_pyramid_scale": [C.c_float, C.c_uint32],
    "zg_pyramid_build_level": [_IMG, _IMG, C.c_float, C.c_void_p],
    "zg_pyramid_level": [C.c_uint32, C.c_uint32, C.c_float, C.c_float, _U32P, _U32P, _F32P],
    "zg_batch_blur_resize": [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_float,
                             C.c_void_p, C.c_uint32, C.c_uint32, _METHOD, C.c_void_p],
    "zg_png_default_limits": [C.POINTER(ZgPngLimits)],
    "zg_png_default_encode_options": [C.POINTER(ZgPngEncodeOptions)],
    "zg_png_info": [C.c_void_p, C.c_size_t, C.POINTER(ZgPngLimits), C.POINTER(ZgPngHeader)],
    "zg_png_probe": [C.c_void_p, C.c_size_t, C.POINTER(ZgPngLimits), C.POINTER(ZgPngHeader), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "zg_png_scan_hash": [C.c_void_p, C.c_size_t, C.POINTER(ZgPngLimits), C.POINTER(C.c_uint64), C.POINTER(C.c_int)],
    "zg_png_decode": [C.c_void_p, C.c_size_t, C.POINTER(ZgPngLimits), _IMG, C.c_int, C.POINTER(C.c_int), C.c_void_p],
    "zg_png_decode_host": [C.c_void_p, C.c_size_t, C.POINTER(ZgPngLimits), _IMG, C.c_int, C.POINTER(C.c_int)],
    "zg_png_filter": [_IMG, C.c_int, C.c_void_p, C.c_void_p],
    "zg_png_encode": [_IMG, C.c_int, C.POINTER(ZgPngEncodeOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p],
    "zg_png_encode_host": [_IMG, C.c_int, C.POINTER(ZgPngEncodeOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
    "zg_png_compress": [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
    "zg_png_free": [C.c_void_p],
    "zg_jpeg_default_limits": [C.POINTER(ZgJpegLimits)],
    "zg_jpeg_info": [C.c_void_p, C.c_size_t, C.POINTER(ZgJpegLimits), C.POINTER(ZgJpegHeader)],
    "zg_jpeg_probe": [C.c_void_p, C.c_size_t, C.POINTER(ZgJpegLimits), C.POINTER(ZgJpegHeader), C.POINTER(C.c_int)],
    "zg_jpeg_coefficient_hash": [C.c_void_p, C.c_size_t, C.POINTER(ZgJpegLimits), C.POINTER(C.c_uint64)],
    "zg_jpeg_decode": [C.c_void_p, C.c_size_t, C.POINTER(ZgJpegLimits), _IMG, C.c_int, C.POINTER(C.c_int), C.c_void_p],
    "zg_jpeg_decode_host": [C.c_void_p, C.c_size_t, C.POINTER(ZgJpegLimits), _IMG, C.c_int, C.POINTER(C.c_int)],
    "zg_jpeg_default_encode_options": [C.POINTER(ZgJpegEncodeOptions)],
    "zg_jpeg_encode": [_IMG, C.c_int, C.POINTER(ZgJpegEncodeOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p],
    "zg_jpeg_encode_host": [_IMG, C.c_int, C.POINTER(ZgJpegEncodeOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
    "zg_jpeg_encode_blocks": [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(ZgJpegEncodeOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
    "zg_jpeg_free": [C.c_void_p],
}
_RESTYPES = {"zg_last_error": C.c_char_p, "zg_shutdown": None, "zg_pixel_size": C.c_size_t, "zg_sizeof_step": C.c_size_t, "zg_pyramid_scale": C.c_float,
             "zg_png_default_limits": None, "zg_png_default_encode_options": None, "zg_png_free": None, "zg_jpeg_default_limits": None, "zg_jpeg_default_encode_options": None, "zg_jpeg_free": None}

# every symbol include/zignal_hip.h declares; tests check the library exports all of them
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). zignal_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library does not export it
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        if l.zg_sizeof_step() != C.sizeof(ZgStep):  # zg_step has no size field: a library built from another header must not be handed ZgStep arrays
            raise ImportError(f"{LIB_PATH}: sizeof(zg_step) is {l.zg_sizeof_step()} in the library, {C.sizeof(ZgStep)} in this binding: rebuild the library")
        _lib = l
    return _lib


def check(status: int) -> None:
    if status == OK:
        return
    msg = (lib().zg_last_error() or b"").decode("utf-8", "replace")
    if status == ERR_DIMENSION_MISMATCH:
        raise DimensionMismatch(status, msg)
    if status == ERR_INVALID_ARGUMENT:
        raise InvalidArgument(status, msg)
    if status == ERR_OUT_OF_MEMORY:
        raise MemoryError(f"zignal_hip: {msg}")
    if status == ERR_CODEC:
        raise CodecError(status, msg)
    raise ZignalError(status, msg)
