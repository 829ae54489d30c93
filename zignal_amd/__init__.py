"""zignal_amd — MI355X (gfx950) implementation of zignal's per-pixel image hot path.

The product is `libzignal_hip.so` (hand-written HIP kernels behind the C ABI in
include/zignal_hip.h); this package is the thin host-side mirror of the reference's `Image(T)`
surface on top of it. There is no CPU fallback: importing works without a GPU, calling does not.
"""
from ._lib import (BORDER_MIRROR, BORDER_REPLICATE, BORDER_WRAP, BORDER_ZERO, CS_GRAY, CS_HSL, CS_HSV, CS_LAB, CS_LCH, CS_LMS, CS_OKLAB, CS_OKLCH, CS_RGB,
                   CS_RGBA, CS_XYB, CS_XYZ, CS_YCBCR, CodecError, DimensionMismatch, InvalidArgument, ZignalError, lib)
from .image import (AffineTransform, Blending, BorderMode, Image, ImagePyramid, Interpolation, ProjectiveTransform,
                    SimilarityTransform, convolve_separable_planes, gaussian_blur_planes, gaussian_kernel, lanczos_plane_weights)
from .pipeline import Multi, Pipeline, Step

__all__ = ["Image", "ImagePyramid", "Interpolation", "BorderMode", "Blending", "ProjectiveTransform", "AffineTransform",
           "SimilarityTransform", "Pipeline", "Step", "Multi", "gaussian_kernel", "gaussian_blur_planes", "convolve_separable_planes", "lanczos_plane_weights", "DimensionMismatch", "InvalidArgument", "CodecError", "ZignalError", "lib", "png", "jpeg"]

from . import jpeg, png  # noqa: E402,F401
