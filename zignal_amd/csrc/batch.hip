// batch.hip — the `pipeline` recipe [blur gaussian sigma, resize bilinear] over a batch of frames
// (semantics of reference src/cli/pipeline.zig:153-179 applied to N independent images laid out back to
// back). Frames are independent units: this is also the shard a rank processes in the multi-GPU bench
// (frame i -> GPU i mod N, no halo, no collective on the data path).
//
// Each frame runs Image.gaussianBlur into a device scratch frame, then Image.resize into its slot of the
// output batch — exactly the two reference calls, so the result equals calling them one after the other.
#include "zg_common.h"

namespace zg {
int resize_impl(const zg_image *src, const zg_image *dst, const zg_method *method, hipStream_t s);
}

using namespace zg;

extern "C" {

int zg_batch_blur_resize(const void *src_frames, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, float sigma,
                         void *dst_frames, uint32_t out_rows, uint32_t out_cols, const zg_method *method, zg_stream stream) {
    ZG_REQUIRE(pixel_valid(pixel), ZG_ERR_INVALID_ARGUMENT, "batch: invalid pixel type %d", pixel);
    ZG_REQUIRE(method != nullptr, ZG_ERR_INVALID_ARGUMENT, "batch: null method");
    if (n_frames == 0 || rows == 0 || cols == 0 || out_rows == 0 || out_cols == 0) return ZG_OK;
    ZG_REQUIRE(src_frames && dst_frames, ZG_ERR_INVALID_ARGUMENT, "batch: null frame pointer");
    hipStream_t s = as_stream(stream);
    const size_t ps = pixel_size(pixel);
    const size_t in_bytes = (size_t)rows * cols * ps, out_bytes = (size_t)out_rows * out_cols * ps;
    // two scratch frames: frame i+1's blur may start while frame i's resize still reads its scratch
    void *scratch = nullptr;
    ZG_HIP(hipMallocAsync(&scratch, 2 * in_bytes, s));
    int rc = ZG_OK;
    for (uint32_t i = 0; i < n_frames && rc == ZG_OK; ++i) {
        zg_image src{(char *)src_frames + (size_t)i * in_bytes, cols, rows, cols, pixel};
        zg_image tmp{(char *)scratch + (size_t)(i & 1) * in_bytes, cols, rows, cols, pixel};
        zg_image dst{(char *)dst_frames + (size_t)i * out_bytes, out_cols, out_rows, out_cols, pixel};
        rc = zg_gaussian_blur(&src, &tmp, sigma, stream);
        if (rc == ZG_OK) rc = resize_impl(&tmp, &dst, method, s);
    }
    (void)hipFreeAsync(scratch, s);
    return rc;
}

} // extern "C"
