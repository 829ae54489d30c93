// zg_hostmath.h — host-side scalar maths used to build the small tables the kernels take as
// arguments: Gaussian taps (reference src/image.zig:973-990, Zig @exp), rotation cos/sin (Zig
// @cos/@sin), the 256-entry sRGB->linear table (std.math.pow, src/color.zig:1252-1258) and the
// 1025-entry Lanczos3 table (src/image/interpolation.zig:256-267).
//
// Zig's std/compiler-rt maths (pinned only by build.zig.zon:5) is not available here; these follow
// the published musl / Go algorithms Zig ports. Every entry point that consumes such a value also
// accepts it from the caller, so a Zig host passes Zig's own numbers and nothing here is on its path.
#pragma once
#include <stdint.h>

namespace zg { namespace hostmath {
float exp_f32(float x);
float log_f32(float x);
float pow_f32(float x, float y);
float sin_f32(float x);
float cos_f32(float x);
float srgb_to_linear(float c);          // gammaToLinear, src/color.zig:1252-1258
const float *srgb_u8_lut();             // [256], gammaToLinear(i / 255)
const float *lanczos3_lut();            // [1025], src/image/interpolation.zig:256-267
}} // namespace zg::hostmath
