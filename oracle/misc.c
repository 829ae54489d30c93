/*
 * oracle/misc.c — Image(T) container helpers. TEST INFRASTRUCTURE ONLY (see zo.h).
 *   src/image.zig:375-392   copy (row-wise, honours views)
 *   src/image.zig:187-230   fill / setBorder
 *   src/image/transforms.zig:28-44  flipLeftRight / flipTopBottom
 */
#include "zo.h"
#include <string.h>

int zo_copy(const zo_image *src, const zo_image *dst) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel) return 2;
    if (src->data == dst->data) return 0;
    const size_t ps = zo_pixel_size(src->pixel);
    for (size_t r = 0; r < src->rows; ++r)
        memcpy((char *)dst->data + r * dst->stride * ps, (const char *)src->data + r * src->stride * ps, src->cols * ps);
    return 0;
}

int zo_fill(const zo_image *img, const void *pixel) {
    const size_t ps = zo_pixel_size(img->pixel);
    for (size_t r = 0; r < img->rows; ++r)
        for (size_t c = 0; c < img->cols; ++c)
            memcpy((char *)img->data + (r * img->stride + c) * ps, pixel, ps);
    return 0;
}

/* setBorder (image.zig:200-230): every pixel outside rect (clipped to the image) gets `pixel`;
 * an empty intersection fills the whole image. rect = {l,t,r,b}, r/b exclusive. */
int zo_set_border(const zo_image *img, const uint32_t rect[4], const void *pixel) {
    const size_t ps = zo_pixel_size(img->pixel);
    uint32_t l = rect[0], t = rect[1], r = rect[2], b = rect[3];
    if (r > img->cols) r = img->cols;
    if (b > img->rows) b = img->rows;
    if (l >= r || t >= b) return zo_fill(img, pixel);
    for (size_t y = 0; y < img->rows; ++y)
        for (size_t x = 0; x < img->cols; ++x)
            if (y < t || y >= b || x < l || x >= r)
                memcpy((char *)img->data + (y * img->stride + x) * ps, pixel, ps);
    return 0;
}

int zo_flip_left_right(const zo_image *img) {
    const size_t ps = zo_pixel_size(img->pixel);
    char tmp[16];
    for (size_t r = 0; r < img->rows; ++r) {
        char *row = (char *)img->data + r * img->stride * ps;
        for (size_t i = 0, j = img->cols; i + 1 < j; ++i) {
            --j;
            if (i >= j) break;
            memcpy(tmp, row + i * ps, ps);
            memcpy(row + i * ps, row + j * ps, ps);
            memcpy(row + j * ps, tmp, ps);
        }
    }
    return 0;
}

int zo_flip_top_bottom(const zo_image *img) {
    const size_t ps = zo_pixel_size(img->pixel);
    char tmp[16];
    for (size_t r = 0; r < img->rows / 2; ++r) {
        char *top = (char *)img->data + r * img->stride * ps;
        char *bot = (char *)img->data + (img->rows - r - 1) * img->stride * ps;
        for (size_t c = 0; c < img->cols; ++c) {
            memcpy(tmp, top + c * ps, ps);
            memcpy(top + c * ps, bot + c * ps, ps);
            memcpy(bot + c * ps, tmp, ps);
        }
    }
    return 0;
}
