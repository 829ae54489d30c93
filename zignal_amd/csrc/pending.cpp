// pending.cpp — entry points declared in include/zignal_hip.h whose kernels have not landed yet.
// They fail loudly with ZG_ERR_UNSUPPORTED; nothing here computes anything. Shrinks to zero.
#include "zg_common.h"
using namespace zg;
#define ZG_PENDING(name, ...) \
    int name(__VA_ARGS__) { set_error(#name ": not implemented yet"); return ZG_ERR_UNSUPPORTED; }
extern "C" {
ZG_PENDING(zg_convolve, const zg_image *, const zg_image *, const float *, uint32_t, uint32_t, int, zg_stream)
ZG_PENDING(zg_convolve_host, const zg_image *, const zg_image *, const float *, uint32_t, uint32_t, int)
ZG_PENDING(zg_box_blur, const zg_image *, const zg_image *, uint32_t, zg_stream)
ZG_PENDING(zg_box_blur_host, const zg_image *, const zg_image *, uint32_t)
ZG_PENDING(zg_resize, const zg_image *, const zg_image *, const zg_method *, zg_stream)
ZG_PENDING(zg_resize_host, const zg_image *, const zg_image *, const zg_method *)
ZG_PENDING(zg_letterbox, const zg_image *, const zg_image *, const zg_method *, uint32_t *, zg_stream)
ZG_PENDING(zg_letterbox_host, const zg_image *, const zg_image *, const zg_method *, uint32_t *)
ZG_PENDING(zg_warp, const zg_image *, const zg_image *, int, const float *, const zg_method *, zg_stream)
ZG_PENDING(zg_warp_host, const zg_image *, const zg_image *, int, const float *, const zg_method *)
ZG_PENDING(zg_rotate_into, const zg_image *, const zg_image *, float, float, float, const zg_method *, int, zg_stream)
ZG_PENDING(zg_rotate_into_host, const zg_image *, const zg_image *, float, float, float, const zg_method *, int)
ZG_PENDING(zg_rotate_bounds, uint32_t, uint32_t, float, float, float, uint32_t *, uint32_t *)
ZG_PENDING(zg_extract, const zg_image *, const zg_image *, const float *, float, float, float, const zg_method *, int, zg_stream)
ZG_PENDING(zg_extract_host, const zg_image *, const zg_image *, const float *, float, float, float, const zg_method *, int)
ZG_PENDING(zg_crop, const zg_image *, const zg_image *, const float *, zg_stream)
ZG_PENDING(zg_crop_host, const zg_image *, const zg_image *, const float *)
ZG_PENDING(zg_crop_dims, const float *, uint32_t *, uint32_t *)
ZG_PENDING(zg_insert, const zg_image *, const zg_image *, const float *, float, float, float, const zg_method *, int, zg_stream)
ZG_PENDING(zg_insert_host, const zg_image *, const zg_image *, const float *, float, float, float, const zg_method *, int)
ZG_PENDING(zg_convert, const zg_image *, int, const zg_image *, int, const float *, zg_stream)
ZG_PENDING(zg_convert_host, const zg_image *, int, const zg_image *, int, const float *)
ZG_PENDING(zg_batch_blur_resize, const void *, uint32_t, uint32_t, uint32_t, int, float, void *, uint32_t, uint32_t, const zg_method *, zg_stream)
}
