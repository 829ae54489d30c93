// zg_bilinear_u8.h — the u8 plane bilinear resizer's per-pixel arithmetic (reference src/image/channel_ops.zig:144-190:
// s = (d + 0.5) * ratio - 0.5 in f32, f = trunc(frac * 256), mirror-resolved taps, two 8.8 lerps and a truncating shift),
// shared by k_resize_bilinear_rgba8 (resize_planes.hip) and the fused resize -> convert kernel (convert.hip).
#pragma once
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

__device__ inline void bilinear_taps(int d, float ratio, int n, int &i0, int &i1, int &f) {
    const float sf = ((float)d + 0.5f) * ratio - 0.5f;
    const float fl = floorf(sf);
    const int base = (int)fl;
    f = (int)truncf((sf - fl) * 256.0f);
    i0 = base;
    i1 = base + 1;
    if (base < 0 || base + 1 >= n) { i0 = resolve_index(base, n, ZG_BORDER_MIRROR); i1 = resolve_index(base + 1, n, ZG_BORDER_MIRROR); }
}
__device__ inline uint32_t bilinear_rgba8(uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br, int fx, int fy) {
    uint32_t px = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const int a = (int)((tl >> (8 * ch)) & 0xffu), b = (int)((tr >> (8 * ch)) & 0xffu);
        const int c = (int)((bl >> (8 * ch)) & 0xffu), d = (int)((br >> (8 * ch)) & 0xffu);
        const int top = a * (256 - fx) + b * fx;
        const int bottom = c * (256 - fx) + d * fx;
        px |= (uint32_t)((top * (256 - fy) + bottom * fy) >> 16) << (8 * ch); // @divTrunc(.., 65536), operand >= 0, <= 255
    }
    return px;
}

} // namespace zg
