// zg_colordev.h — the Rgb -> Xyz -> Oklab arithmetic of convertColor (reference src/color.zig:1261-1272 rgbToXyz after
// gammaToLinear, :1381-1400 xyzToOklab) as device functions, shared by k_convert (convert.hip) and the fused
// resize -> convert kernel (resize_planes.hip) so that both produce the same bits by construction.
#pragma once
#include "zg_devmath.h"

#pragma clang fp contract(off)

namespace zg {

// linear RGB -> Xyz scaled by 100 (color.zig:1261-1272)
__device__ inline void linear_rgb_to_xyz(const float lin[3], float &X, float &Y, float &Z) {
    X = (lin[0] * 0.4124f + lin[1] * 0.3576f + lin[2] * 0.1805f) * 100;
    Y = (lin[0] * 0.2126f + lin[1] * 0.7152f + lin[2] * 0.0722f) * 100;
    Z = (lin[0] * 0.0193f + lin[1] * 0.1192f + lin[2] * 0.9505f) * 100;
}

// Xyz -> Oklab (color.zig:1381-1400)
__device__ inline void xyz_to_oklab(float X, float Y, float Z, float &L, float &A, float &B) {
    const float x = dev_div100(X), y = dev_div100(Y), z = dev_div100(Z); // == X / 100.0f, bit for bit
    const float l_linear = 0.8189330101f * x + 0.3618667424f * y - 0.1288597137f * z;
    const float m_linear = 0.0329845436f * x + 0.9293118715f * y + 0.0361456387f * z;
    const float s_linear = 0.0482003018f * x + 0.2643662691f * y + 0.6338517070f * z;
    const float l_dash = dev_cbrtf(l_linear), m_dash = dev_cbrtf(m_linear), s_dash = dev_cbrtf(s_linear);
    L = 0.2104542553f * l_dash + 0.7936177850f * m_dash - 0.0040720468f * s_dash;
    A = 1.9779984951f * l_dash - 2.4285922050f * m_dash + 0.4505937099f * s_dash;
    B = 0.0259040371f * l_dash + 0.7827717662f * m_dash - 0.8086757660f * s_dash;
}

} // namespace zg
