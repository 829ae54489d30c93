//! zignal_hip.zig — the Zig side of the drop-in: `Image(T)` whose hot-path methods call libzignal_hip.so.
//!
//! NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Zig toolchain (the reference needs Zig
//! >= 0.17.0-dev.1441, build.zig.zon:5). It is kept thin on purpose — extern declarations of the C ABI in
//! include/zignal_hip.h plus method bodies that forward — and mirrors, signature for signature, the methods
//! of reference src/image.zig that it replaces (line numbers cited per method). Everything that depends on
//! Zig's own maths (@exp for Gaussian taps, @cos/@sin for rotations, std.math.pow for the sRGB table, @sin
//! for the Lanczos table) is computed HERE, in Zig, and handed to the library as plain numbers, so results
//! keep Zig's bit patterns.
//!
//! Use: in build.zig add `exe.linkSystemLibrary("zignal_hip")` (+ library path), and
//! `const Image = @import("zignal_hip.zig").Image;` in place of zignal's Image for the hot path.
const std = @import("std");
const zignal = @import("zignal");

pub const BorderMode = zignal.BorderMode; // ordinals == zg_border
pub const Interpolation = zignal.Interpolation; // tag ordinals == zg_interp
const Rectangle = zignal.Rectangle;
const Point = zignal.Point;

// ---- C ABI (include/zignal_hip.h) -----------------------------------------------------------------
/// The `pipeline` command (src/cli/pipeline.zig:153-179) over a batch of equally shaped frames resident on the device: every frame
/// goes through `steps` in order with one zg_batch_pipeline call (a launch per step over the whole batch where the library has a
/// batched kernel, fused neighbours where it has a fused one). `src` / `dst` are device pointers (DeviceImage memory, zg_malloc).
pub const Pipeline = struct {
    steps: []const c.ZgStep,

    pub fn gaussianBlur(sigma: f32) c.ZgStep {
        return .{ .kind = 0, .sigma = sigma, .radius = 0, .out_rows = 0, .out_cols = 0, .method = .{ .kind = 0, .b = 0, .c = 0, .lanczos_lut = null }, .dst_pixel = 0, .dst_space = 0, .srgb_lut = null, .transform = 0, .m = [_]f32{0} ** 9, .motion = 0, .angle = 0, .cos_a = 1, .sin_a = 0, .distance = 0, .center_x = 0.5, .center_y = 0.5, .strength = 0.5, .edges = 0, .low = 0, .high = 0, .window = 7, .use_nms = 0 };
    }
    /// blur --type median (src/cli/blur.zig:116-123)
    pub fn medianBlur(radius: u32) c.ZgStep {
        var s = gaussianBlur(0);
        s.kind = 5;
        s.radius = radius;
        return s;
    }
    /// blur --type motion_linear (src/cli/blur.zig:124-146): angle in radians; the cosine and sine are Zig's own
    pub fn motionBlurLinear(angle: f32, distance: u32) c.ZgStep {
        var s = gaussianBlur(0);
        s.kind = 6;
        s.motion = 0;
        s.angle = angle;
        s.cos_a = @cos(angle);
        s.sin_a = @sin(angle);
        s.distance = distance;
        return s;
    }
    /// blur --type motion_zoom / motion_spin (src/cli/blur.zig:147-170)
    pub fn motionBlurRadial(center_x: f32, center_y: f32, strength: f32, spin: bool) c.ZgStep {
        var s = gaussianBlur(0);
        s.kind = 6;
        s.motion = if (spin) 2 else 1;
        s.center_x = center_x;
        s.center_y = center_y;
        s.strength = strength;
        return s;
    }
    /// edges --filter sobel through edges.apply's grey bridge (src/cli/edges.zig:126-135); the frames keep their type
    pub fn edgesSobel() c.ZgStep {
        var s = gaussianBlur(0);
        s.kind = 7;
        s.edges = 0;
        return s;
    }
    pub fn edgesCanny(sigma: f32, low: f32, high: f32) c.ZgStep {
        var s = gaussianBlur(sigma);
        s.kind = 7;
        s.edges = 1;
        s.low = low;
        s.high = high;
        return s;
    }
    pub fn edgesShenCastan(opts: zignal.ShenCastan) c.ZgStep {
        var s = gaussianBlur(opts.smooth);
        s.kind = 7;
        s.edges = 2;
        s.window = @intCast(opts.window_size);
        s.high = opts.high_ratio;
        s.low = opts.low_rel;
        s.use_nms = @intFromBool(opts.use_nms);
        return s;
    }
    pub fn boxBlur(radius: u32) c.ZgStep {
        var s = gaussianBlur(0);
        s.kind = 1;
        s.radius = radius;
        return s;
    }
    pub fn resize(rows: u32, cols: u32, method: c.ZgMethod) c.ZgStep {
        var s = gaussianBlur(0);
        s.kind = 2;
        s.out_rows = rows;
        s.out_cols = cols;
        s.method = method;
        return s;
    }
    pub fn convert(dst_pixel: c_int, dst_space: c_int) c.ZgStep {
        var s = gaussianBlur(0);
        s.kind = 3;
        s.dst_pixel = dst_pixel;
        s.dst_space = dst_space;
        return s;
    }
    /// zg_step carries no size field: a library built from another header must not be handed arrays of this file's ZgStep (zignal_hip.h: zg_sizeof_step)
    fn checkStepAbi() !void {
        if (c.zg_sizeof_step() != @sizeOf(c.ZgStep)) return error.AbiMismatch;
    }
    /// rows, cols, pixel type and colour space of the frames after the steps
    pub fn outShape(self: Pipeline, rows: u32, cols: u32, pixel: c_int, space: c_int) !struct { rows: u32, cols: u32, pixel: c_int, space: c_int } {
        try checkStepAbi();
        var r: u32 = 0;
        var cc: u32 = 0;
        var p: c_int = 0;
        var sp: c_int = 0;
        try check(c.zg_batch_pipeline_shape(rows, cols, pixel, space, self.steps.ptr, @intCast(self.steps.len), &r, &cc, &p, &sp));
        return .{ .rows = r, .cols = cc, .pixel = p, .space = sp };
    }
    pub fn run(self: Pipeline, src: *const anyopaque, n_frames: u32, rows: u32, cols: u32, pixel: c_int, space: c_int, dst: *anyopaque, stream: ?*anyopaque) !void {
        try checkStepAbi();
        try check(c.zg_batch_pipeline(src, n_frames, rows, cols, pixel, space, self.steps.ptr, @intCast(self.steps.len), dst, stream));
    }
    /// The same over every device of a zg_multi context (zg_multi_batch_pipeline): src / dst live on the context's root device; frames shard in
    /// contiguous blocks, no halo, no collective on the data path. Synchronous.
    pub fn runMulti(self: Pipeline, ctx: ?*anyopaque, src_root: *const anyopaque, n_frames: u32, rows: u32, cols: u32, pixel: c_int, space: c_int, dst_root: *anyopaque, times_ms: ?*[3]f32) !void {
        try checkStepAbi();
        try check(c.zg_multi_batch_pipeline(ctx, src_root, n_frames, rows, cols, pixel, space, self.steps.ptr, @intCast(self.steps.len), dst_root, times_ms));
    }
};

pub const c = struct {
    pub const ZgImage = extern struct { data: ?*anyopaque, stride: usize, rows: u32, cols: u32, pixel: i32 };
    pub const ZgMethod = extern struct { kind: i32, b: f32, c: f32, lanczos_lut: ?[*]const f32 };
    /// zg_step: one step of zg_batch_pipeline (kind: 0 gaussian blur, 1 box blur, 2 resize, 3 convert, 4 warp, 5 median blur, 6 motion blur, 7 edges)
    pub const ZgStep = extern struct { kind: c_int, sigma: f32, radius: u32, out_rows: u32, out_cols: u32, method: ZgMethod, dst_pixel: c_int, dst_space: c_int, srgb_lut: ?[*]const f32, transform: c_int, m: [9]f32, motion: c_int, angle: f32, cos_a: f32, sin_a: f32, distance: u32, center_x: f32, center_y: f32, strength: f32, edges: c_int, low: f32, high: f32, window: u32, use_nms: c_int };
    pub extern fn zg_init(device: c_int) c_int;
    pub extern fn zg_last_error() [*:0]const u8;
    pub extern fn zg_conv_separable_host(src: *const ZgImage, dst: *const ZgImage, kx: [*]const f32, nkx: u32, ky: [*]const f32, nky: u32, border: c_int) c_int;
    pub extern fn zg_convolve_host(src: *const ZgImage, dst: *const ZgImage, kernel: [*]const f32, kh: u32, kw: u32, border: c_int) c_int;
    pub extern fn zg_box_blur_host(src: *const ZgImage, dst: *const ZgImage, radius: u32) c_int;
    pub extern fn zg_resize_host(src: *const ZgImage, dst: *const ZgImage, method: *const ZgMethod) c_int;
    pub extern fn zg_letterbox_host(src: *const ZgImage, dst: *const ZgImage, method: *const ZgMethod, rect_out: *[4]u32) c_int;
    pub extern fn zg_warp_host(src: *const ZgImage, dst: *const ZgImage, kind: c_int, m: [*]const f32, method: *const ZgMethod) c_int;
    pub extern fn zg_rotate_into_host(src: *const ZgImage, dst: *const ZgImage, angle: f32, cos_a: f32, sin_a: f32, method: *const ZgMethod, border: c_int) c_int;
    pub extern fn zg_extract_host(src: *const ZgImage, dst: *const ZgImage, rect: *const [4]f32, angle: f32, cos_a: f32, sin_a: f32, method: *const ZgMethod, border: c_int) c_int;
    pub extern fn zg_insert_host(self: *const ZgImage, source: *const ZgImage, rect: *const [4]f32, angle: f32, cos_a: f32, sin_a: f32, method: *const ZgMethod, blend_mode: c_int) c_int;
    pub extern fn zg_fill_host(img: *const ZgImage, pixel_value: *const anyopaque) c_int;
    pub extern fn zg_set_border_host(img: *const ZgImage, rect: *const [4]u32, pixel_value: *const anyopaque) c_int;
    pub extern fn zg_crop_host(src: *const ZgImage, dst: *const ZgImage, rect: *const [4]f32) c_int;
    pub extern fn zg_crop_dims(rect: *const [4]f32, out_rows: *u32, out_cols: *u32) c_int;
    pub extern fn zg_gaussian_blur_host(src: *const ZgImage, dst: *const ZgImage, sigma: f32) c_int;
    pub extern fn zg_gaussian_kernel(sigma: f32, taps: [*]f32, capacity: u32) c_int;
    pub extern fn zg_rotate_bounds(rows: u32, cols: u32, angle: f32, cos_a: f32, sin_a: f32, out_rows: *u32, out_cols: *u32) c_int;
    pub extern fn zg_pyramid_scale(scale_factor: f32, level: u32) f32;
    pub extern fn zg_pyramid_level(rows: u32, cols: u32, scale: f32, blur_sigma: f32, out_rows: *u32, out_cols: *u32, out_sigma: *f32) c_int;
    pub extern fn zg_png_scan_hash(png: [*]const u8, len: usize, limits: ?*const ZgPngLimits, hash_out: *u64, truncated_out: ?*c_int) c_int;
    pub extern fn zg_jpeg_coefficient_hash(jpeg: [*]const u8, len: usize, limits: ?*const ZgJpegLimits, hash_out: *u64) c_int;
    // Device-resident frames (pointers from zg_malloc, work ordered on a zg_stream): what a pipeline that keeps its images in
    // HBM between calls uses instead of the *_host entry points. Every *_host function above has the same-named twin
    // without the suffix and with a trailing `stream` argument (include/zignal_hip.h); these are the ones around them.
    pub extern fn zg_version() c_int;
    pub extern fn zg_device_count() c_int;
    pub extern fn zg_shutdown() void;
    pub extern fn zg_pixel_size(pixel: c_int) usize;
    pub extern fn zg_malloc(dev_ptr: *?*anyopaque, bytes: usize) c_int;
    pub extern fn zg_free(dev_ptr: ?*anyopaque) c_int;
    pub extern fn zg_memcpy_h2d(dst_dev: *anyopaque, src_host: *const anyopaque, bytes: usize, stream: ?*anyopaque) c_int;
    pub extern fn zg_memcpy_d2h(dst_host: *anyopaque, src_dev: *const anyopaque, bytes: usize, stream: ?*anyopaque) c_int;
    pub extern fn zg_set_device(device: c_int) c_int;
    pub extern fn zg_get_device(device: *c_int) c_int;
    pub extern fn zg_malloc_host(host_ptr: *?*anyopaque, bytes: usize) c_int;
    pub extern fn zg_free_host(host_ptr: ?*anyopaque) c_int;
    pub extern fn zg_memcpy_h2d_async(dst_dev: *anyopaque, src_host: *const anyopaque, bytes: usize, stream: ?*anyopaque) c_int;
    pub extern fn zg_memcpy_d2h_async(dst_host: *anyopaque, src_dev: *const anyopaque, bytes: usize, stream: ?*anyopaque) c_int;
    pub extern fn zg_image_upload(dst_dev: *const ZgImage, src_host: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_image_download(dst_host: *const ZgImage, src_dev: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_stream_wait_event(stream: ?*anyopaque, event: ?*anyopaque) c_int;
    pub extern fn zg_event_create(out: *?*anyopaque) c_int;
    pub extern fn zg_event_destroy(event: ?*anyopaque) c_int;
    pub extern fn zg_event_record(event: ?*anyopaque, stream: ?*anyopaque) c_int;
    pub extern fn zg_event_synchronize(event: ?*anyopaque) c_int;
    pub extern fn zg_event_elapsed_ms(start: ?*anyopaque, stop: ?*anyopaque, ms: *f32) c_int;
    pub extern fn zg_graph_begin_capture(stream: ?*anyopaque) c_int;
    pub extern fn zg_graph_end_capture(stream: ?*anyopaque, out: *?*anyopaque) c_int;
    pub extern fn zg_graph_launch(graph: ?*anyopaque, stream: ?*anyopaque) c_int;
    pub extern fn zg_graph_destroy(graph: ?*anyopaque) c_int;
    pub extern fn zg_release_graph_scratch() c_int;
    pub extern fn zg_trim_scratch() c_int;
    pub extern fn zg_devmath_apply(func: c_int, x_dev: [*]const f32, y_dev: ?[*]const f32, out_dev: [*]f32, n: usize, stream: ?*anyopaque) c_int;
    pub extern fn zg_lanczos_plane_weights(src_n: u32, dst_n: u32, weights: [*]f32) c_int;
    pub extern fn zg_resize_lanczos_weights(src: *const ZgImage, dst: *const ZgImage, wx: ?[*]const f32, wy: ?[*]const f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_resize_lanczos_weights_host(src: *const ZgImage, dst: *const ZgImage, wx: ?[*]const f32, wy: ?[*]const f32) c_int;
    pub extern fn zg_resize_convert(src: *const ZgImage, src_space: c_int, dst: *const ZgImage, dst_space: c_int, method: *const ZgMethod, srgb_lut: ?[*]const f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_resize_convert_host(src: *const ZgImage, src_space: c_int, dst: *const ZgImage, dst_space: c_int, method: *const ZgMethod, srgb_lut: ?[*]const f32) c_int;
    pub extern fn zg_batch_pipeline_shape(rows: u32, cols: u32, pixel: c_int, space: c_int, steps: ?[*]const ZgStep, n_steps: u32, out_rows: ?*u32, out_cols: ?*u32, out_pixel: ?*c_int, out_space: ?*c_int) c_int;
    pub extern fn zg_batch_pipeline(src_frames: ?*const anyopaque, n_frames: u32, rows: u32, cols: u32, pixel: c_int, space: c_int, steps: ?[*]const ZgStep, n_steps: u32, dst_frames: ?*anyopaque, stream: ?*anyopaque) c_int;
    pub extern fn zg_pyramid_build(source: *const ZgImage, levels: ?[*]const ZgImage, sigmas: ?[*]const f32, n_levels: u32, stream: ?*anyopaque) c_int;
    pub extern fn zg_multi_create(devices: ?[*]const c_int, n_devices: c_int, out: *?*anyopaque) c_int;
    pub extern fn zg_multi_destroy(m: ?*anyopaque) c_int;
    pub extern fn zg_multi_device_count(m: ?*anyopaque) c_int;
    pub extern fn zg_multi_wait_stream(m: ?*anyopaque, producer: ?*anyopaque) c_int;
    pub extern fn zg_multi_batch_pipeline(m: ?*anyopaque, src_frames_root: *const anyopaque, n_frames: u32, rows: u32, cols: u32, pixel: c_int, space: c_int, steps: [*]const ZgStep, n_steps: u32, dst_frames_root: *anyopaque, times_ms: ?*[3]f32) c_int;
    pub extern fn zg_multi_piece_range(n_frames: u32, world: c_int, chunks: c_int, device: c_int, piece: c_int, begin: *u32, end: *u32) c_int;
    pub extern fn zg_sizeof_step() usize;
    pub extern fn zg_multi_batch_blur_resize(m: ?*anyopaque, src_frames_root: *const anyopaque, n_frames: u32, rows: u32, cols: u32, pixel: c_int, sigma: f32, dst_frames_root: *anyopaque, out_rows: u32, out_cols: u32, method: *const ZgMethod, times_ms: ?*[3]f32) c_int;
    pub extern fn zg_stream_create(out: *?*anyopaque) c_int;
    pub extern fn zg_stream_destroy(stream: ?*anyopaque) c_int;
    pub extern fn zg_stream_synchronize(stream: ?*anyopaque) c_int;
    pub extern fn zg_copy(src: *const ZgImage, dst: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_fill(img: *const ZgImage, pixel_value: *const anyopaque, stream: ?*anyopaque) c_int;
    pub extern fn zg_set_border(img: *const ZgImage, rect: *const [4]u32, pixel_value: *const anyopaque, stream: ?*anyopaque) c_int;
    pub extern fn zg_crop(src: *const ZgImage, dst: *const ZgImage, rect: *const [4]f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_gaussian_blur(src: *const ZgImage, dst: *const ZgImage, sigma: f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_conv_separable_planes(src: [*]const ZgImage, dst: [*]const ZgImage, n_planes: u32, kx: [*]const f32, nkx: u32, ky: [*]const f32, nky: u32, border: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_gaussian_blur_planes(src: [*]const ZgImage, dst: [*]const ZgImage, n_planes: u32, sigma: f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_autocontrast(img: *const ZgImage, cutoff: f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_box_blur(src: *const ZgImage, dst: *const ZgImage, radius: u32, stream: ?*anyopaque) c_int;
    pub extern fn zg_canny(src: *const ZgImage, dst: *const ZgImage, sigma: f32, low_threshold: f32, high_threshold: f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_conv_separable(src: *const ZgImage, dst: *const ZgImage, kx: [*]const f32, nkx: u32, ky: [*]const f32, nky: u32, border: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_convert(src: *const ZgImage, src_space: c_int, dst: *const ZgImage, dst_space: c_int, srgb_lut: ?[*]const f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_convolve(src: *const ZgImage, dst: *const ZgImage, kernel: [*]const f32, kh: u32, kw: u32, border: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_equalize(img: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_extract(src: *const ZgImage, dst: *const ZgImage, rect: *const [4]f32, angle: f32, cos_a: f32, sin_a: f32, method: *const ZgMethod, border: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_flip_left_right(img: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_flip_top_bottom(img: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_insert(self: *const ZgImage, source: *const ZgImage, rect: *const [4]f32, angle: f32, cos_a: f32, sin_a: f32, method: *const ZgMethod, blend_mode: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_integral(src: *const ZgImage, planes: [*]f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_invert(img: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_letterbox(src: *const ZgImage, dst: *const ZgImage, method: *const ZgMethod, rect_out: *[4]u32, stream: ?*anyopaque) c_int;
    pub extern fn zg_morph(src: *const ZgImage, dst: *const ZgImage, kernel: [*]const u8, kernel_rows: u32, kernel_cols: u32, iterations: u32, op: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_motion_blur_linear(src: *const ZgImage, dst: *const ZgImage, angle: f32, cos_a: f32, sin_a: f32, distance: u32, stream: ?*anyopaque) c_int;
    pub extern fn zg_motion_blur_radial(src: *const ZgImage, dst: *const ZgImage, center_x: f32, center_y: f32, strength: f32, spin: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_order_statistic_blur(src: *const ZgImage, dst: *const ZgImage, radius: u32, op: c_int, param: f64, border: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_resize(src: *const ZgImage, dst: *const ZgImage, method: *const ZgMethod, stream: ?*anyopaque) c_int;
    pub extern fn zg_rotate_into(src: *const ZgImage, dst: *const ZgImage, angle: f32, cos_a: f32, sin_a: f32, method: *const ZgMethod, border: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_sharpen(src: *const ZgImage, dst: *const ZgImage, radius: u32, stream: ?*anyopaque) c_int;
    pub extern fn zg_isef_smooth(src: *const ZgImage, dst: *const ZgImage, smooth: f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_shen_castan(src: *const ZgImage, dst: *const ZgImage, smooth: f32, window_size: u32, high_ratio: f32, low_rel: f32, hysteresis: c_int, use_nms: c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_sobel(src: *const ZgImage, dst: *const ZgImage, stream: ?*anyopaque) c_int;
    pub extern fn zg_threshold_adaptive_mean(src: *const ZgImage, dst: *const ZgImage, radius: u32, c: f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_threshold_otsu(src: *const ZgImage, dst: *const ZgImage, threshold_out: ?*u8, stream: ?*anyopaque) c_int;
    pub extern fn zg_warp(src: *const ZgImage, dst: *const ZgImage, kind: c_int, m: [*]const f32, method: *const ZgMethod, stream: ?*anyopaque) c_int;
    pub extern fn zg_batch_blur_resize(src_frames: *const anyopaque, n_frames: u32, rows: u32, cols: u32, pixel: c_int, sigma: f32, dst_frames: *anyopaque, out_rows: u32, out_cols: u32, method: *const ZgMethod, stream: ?*anyopaque) c_int;
    pub extern fn zg_flip_left_right_host(img: *const ZgImage) c_int;
    pub extern fn zg_flip_top_bottom_host(img: *const ZgImage) c_int;
    /// ImagePyramid.build's loop body (src/image/pyramid.zig:76-92) for device-resident images: blur when sigma > 0.5, then bilinear resize
    pub extern fn zg_pyramid_build_level(source: *const ZgImage, level: *const ZgImage, sigma: f32, stream: ?*anyopaque) c_int;
    pub extern fn zg_order_statistic_blur_host(src: *const ZgImage, dst: *const ZgImage, radius: u32, op: c_int, param: f64, border: c_int) c_int;
    pub extern fn zg_autocontrast_host(img: *const ZgImage, cutoff: f32) c_int;
    pub extern fn zg_equalize_host(img: *const ZgImage) c_int;
    pub extern fn zg_threshold_otsu_host(src: *const ZgImage, dst: *const ZgImage, threshold_out: ?*u8) c_int;
    pub extern fn zg_threshold_adaptive_mean_host(src: *const ZgImage, dst: *const ZgImage, radius: u32, c: f32) c_int;
    pub extern fn zg_morph_host(src: *const ZgImage, dst: *const ZgImage, kernel: [*]const u8, kernel_rows: u32, kernel_cols: u32, iterations: u32, op: c_int) c_int;
    pub extern fn zg_sharpen_host(src: *const ZgImage, dst: *const ZgImage, radius: u32) c_int;
    pub extern fn zg_integral_host(src: *const ZgImage, planes: [*]f32) c_int;
    pub extern fn zg_invert_host(img: *const ZgImage) c_int;
    pub extern fn zg_sobel_host(src: *const ZgImage, dst: *const ZgImage) c_int;
    pub extern fn zg_motion_blur_linear_host(src: *const ZgImage, dst: *const ZgImage, angle: f32, cos_a: f32, sin_a: f32, distance: u32) c_int;
    pub extern fn zg_motion_blur_radial_host(src: *const ZgImage, dst: *const ZgImage, center_x: f32, center_y: f32, strength: f32, spin: c_int) c_int;
    pub const ZgPngHeader = extern struct { width: u32, height: u32, bit_depth: u8, color_type: u8, compression_method: u8, filter_method: u8, interlace_method: u8, has_gamma: u8, has_srgb: u8, srgb_intent: u8, gamma: f32 };
    pub const ZgPngLimits = extern struct { max_png_bytes: usize, max_chunk_bytes: usize, max_idat_bytes: usize, max_chunks: usize, max_width: u32, max_height: u32, max_pixels: u64, max_decompressed_bytes: usize };
    pub const ZgPngEncodeOptions = extern struct { filter: c_int, compression_level: c_int, has_gamma: c_int, gamma: f32, srgb_intent: c_int };
    pub extern fn zg_png_default_limits(limits: *ZgPngLimits) void;
    pub extern fn zg_png_default_encode_options(options: *ZgPngEncodeOptions) void;
    pub extern fn zg_png_info(png: [*]const u8, len: usize, limits: ?*const ZgPngLimits, out: *ZgPngHeader) c_int;
    pub extern fn zg_png_probe(png: [*]const u8, len: usize, limits: ?*const ZgPngLimits, header_out: ?*ZgPngHeader, native_pixel_out: ?*c_int, truncated_out: ?*c_int) c_int;
    pub extern fn zg_png_decode(png: [*]const u8, len: usize, limits: ?*const ZgPngLimits, dst: *const ZgImage, dst_space: c_int, truncated_out: ?*c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_png_decode_host(png: [*]const u8, len: usize, limits: ?*const ZgPngLimits, dst: *const ZgImage, dst_space: c_int, truncated_out: ?*c_int) c_int;
    pub extern fn zg_png_filter(src: *const ZgImage, filter: c_int, filtered: [*]u8, stream: ?*anyopaque) c_int;
    pub extern fn zg_png_encode(src: *const ZgImage, src_space: c_int, options: ?*const ZgPngEncodeOptions, out: *?[*]u8, out_len: *usize, stream: ?*anyopaque) c_int;
    pub extern fn zg_png_encode_host(src: *const ZgImage, src_space: c_int, options: ?*const ZgPngEncodeOptions, out: *?[*]u8, out_len: *usize) c_int;
    pub extern fn zg_png_compress(scanlines: [*]const u8, len: usize, compression_level: c_int, out: *?[*]u8, out_len: *usize) c_int;
    pub extern fn zg_png_free(p: ?*anyopaque) void;
    pub const ZgJpegHeader = extern struct { width: u32, height: u32, precision: u8, num_components: u8, progressive: u8, subsampling: i8 };
    pub const ZgJpegLimits = extern struct { max_jpeg_bytes: usize, max_marker_bytes: usize, max_width: u32, max_height: u32, max_pixels: u64, max_blocks: usize, max_scans: usize };
    pub extern fn zg_jpeg_default_limits(limits: *ZgJpegLimits) void;
    pub extern fn zg_jpeg_info(jpeg: [*]const u8, len: usize, limits: ?*const ZgJpegLimits, out: *ZgJpegHeader) c_int;
    pub extern fn zg_jpeg_probe(jpeg: [*]const u8, len: usize, limits: ?*const ZgJpegLimits, header_out: ?*ZgJpegHeader, scan_limit_reached_out: ?*c_int) c_int;
    pub extern fn zg_jpeg_decode(jpeg: [*]const u8, len: usize, limits: ?*const ZgJpegLimits, dst: *const ZgImage, dst_space: c_int, scan_limit_reached_out: ?*c_int, stream: ?*anyopaque) c_int;
    pub extern fn zg_jpeg_decode_host(jpeg: [*]const u8, len: usize, limits: ?*const ZgJpegLimits, dst: *const ZgImage, dst_space: c_int, scan_limit_reached_out: ?*c_int) c_int;
    pub const ZgJpegEncodeOptions = extern struct { quality: c_int, subsampling: c_int, density_dpi: c_int, comment: ?[*]const u8, comment_len: usize };
    pub extern fn zg_jpeg_default_encode_options(options: *ZgJpegEncodeOptions) void;
    pub extern fn zg_jpeg_encode(src: *const ZgImage, src_space: c_int, options: ?*const ZgJpegEncodeOptions, out: *?[*]u8, out_len: *usize, stream: ?*anyopaque) c_int;
    pub extern fn zg_jpeg_encode_host(src: *const ZgImage, src_space: c_int, options: ?*const ZgJpegEncodeOptions, out: *?[*]u8, out_len: *usize) c_int;
    pub extern fn zg_jpeg_encode_blocks(blocks: [*]const i16, rows: u32, cols: u32, gray: c_int, options: ?*const ZgJpegEncodeOptions, out: *?[*]u8, out_len: *usize) c_int;
    pub extern fn zg_jpeg_free(p: ?*anyopaque) void;
    pub extern fn zg_shen_castan_host(src: *const ZgImage, dst: *const ZgImage, smooth: f32, window_size: u32, high_ratio: f32, low_rel: f32, hysteresis: c_int, use_nms: c_int) c_int;
    pub extern fn zg_canny_host(src: *const ZgImage, dst: *const ZgImage, sigma: f32, low_threshold: f32, high_threshold: f32) c_int;
    pub extern fn zg_convert_host(src: *const ZgImage, src_space: c_int, dst: *const ZgImage, dst_space: c_int, srgb_lut: ?[*]const f32) c_int;
};

pub const MotionBlur = zignal.MotionBlur; // the reference's union, re-exported

pub const Pixel = enum(i32) { u8 = 0, f32 = 1, rgb_u8 = 2, rgba_u8 = 3, rgb_f32 = 4, rgba_f32 = 5 };

fn pixelOf(comptime T: type) Pixel {
    return switch (T) {
        u8, zignal.Gray(u8) => .u8,
        f32, zignal.Gray(f32) => .f32,
        zignal.Rgb(u8) => .rgb_u8,
        zignal.Rgba(u8) => .rgba_u8,
        zignal.Rgb(f32), zignal.Oklab(f32), zignal.Xyz(f32) => .rgb_f32,
        zignal.Rgba(f32) => .rgba_f32,
        else => @compileError("zignal_hip: pixel type " ++ @typeName(T) ++ " is not on the GPU hot path"),
    };
}

fn check(status: c_int) !void {
    return switch (status) {
        0 => {},
        1 => error.DimensionMismatch,
        2 => error.InvalidArgument,
        3 => error.OutOfMemory,
        else => error.HipFailure,
    };
}

fn methodOf(m: Interpolation, lut: ?[*]const f32) c.ZgMethod {
    return switch (m) {
        .mitchell => |p| .{ .kind = 4, .b = p.b, .c = p.c, .lanczos_lut = null },
        .lanczos => .{ .kind = 5, .b = 0, .c = 0, .lanczos_lut = lut },
        else => .{ .kind = @intFromEnum(std.meta.activeTag(m)), .b = 0, .c = 0, .lanczos_lut = null },
    };
}

/// gammaToLinear(i / 255) for all 256 levels with Zig's own std.math.pow (reference src/color.zig:1252-1258).
fn srgbLut() [256]f32 {
    var lut: [256]f32 = undefined;
    for (&lut, 0..) |*v, i| {
        const cc = @as(f32, @floatFromInt(i)) / 255;
        v.* = if (cc > 0.04045) std.math.pow(f32, (cc + 0.055) / 1.055, 2.4) else cc / 12.92;
    }
    return lut;
}

/// lanczosKernel of resizePlaneLanczosU8 (reference src/image/channel_ops.zig:446-454), evaluated with Zig's own @sin.
fn lanczosPlaneKernel(x: f32) f32 {
    if (x == 0) return 1.0;
    const a = 3.0;
    if (@abs(x) >= a) return 0.0;
    const pi_x = std.math.pi * x;
    return (a * @sin(pi_x) * @sin(pi_x / a)) / (pi_x * pi_x);
}
/// The six plane weights of every destination index of one axis (channel_ops.zig:456-466), for zg_resize_lanczos_weights.
fn lanczosPlaneWeights(allocator: std.mem.Allocator, src_n: u32, dst_n: u32) ![]f32 {
    const w = try allocator.alloc(f32, @as(usize, dst_n) * 6);
    const ratio = @as(f32, @floatFromInt(src_n)) / @as(f32, @floatFromInt(dst_n));
    for (0..dst_n) |d| {
        const s = (@as(f32, @floatFromInt(d)) + 0.5) * ratio - 0.5;
        const f = s - @floor(s);
        for (0..6) |k| w[d * 6 + k] = lanczosPlaneKernel(@as(f32, @floatFromInt(@as(isize, @intCast(k)) - 2)) - f);
    }
    return w;
}
fn isRgbU8(comptime T: type) bool {
    return T == zignal.Rgb(u8) or T == zignal.Rgba(u8);
}

/// Drop-in for zignal.Image(T) on the hot path. Fields and non-hot-path methods are zignal's own.
pub fn Image(comptime T: type) type {
    return struct {
        const Self = @This();
        const Base = zignal.Image(T);
        base: Base,

        fn desc(img: Base) c.ZgImage {
            return .{ .data = @ptrCast(img.data.ptr), .stride = img.stride, .rows = img.rows, .cols = img.cols, .pixel = @intFromEnum(pixelOf(T)) };
        }

        /// reference src/image.zig:935-951
        pub fn convolveSeparable(self: Self, out: Self, allocator: std.mem.Allocator, kernel_x: []const f32, kernel_y: []const f32, border: BorderMode) !void {
            _ = allocator; // scratch lives on the device
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            try check(c.zg_conv_separable_host(&desc(self.base), &desc(out.base), kernel_x.ptr, @intCast(kernel_x.len), kernel_y.ptr, @intCast(kernel_y.len), @intFromEnum(border)));
        }

        /// reference src/image.zig:954-994 — taps built here with Zig's @exp, then the separable kernel.
        pub fn gaussianBlur(self: Self, out: Self, allocator: std.mem.Allocator, sigma: f32) !void {
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            if (sigma == 0) return self.base.copy(out.base);
            if (sigma < 0) return error.InvalidSigma;
            const radius: usize = @ceil(3.0 * sigma);
            const kernel = try allocator.alloc(f32, 2 * radius + 1);
            defer allocator.free(kernel);
            var sum: f32 = 0;
            for (kernel, 0..) |*k, i| {
                const x = @as(f32, @floatFromInt(i)) - @as(f32, @floatFromInt(radius));
                k.* = @exp(-(x * x) / (2.0 * sigma * sigma));
                sum += k.*;
            }
            for (kernel) |*k| k.* /= sum;
            try self.convolveSeparable(out, allocator, kernel, kernel, .mirror);
        }

        /// reference src/image.zig:917-932 — `kernel` is a comptime-sized 2-D array as in the reference.
        pub fn convolve(self: Self, out: Self, allocator: std.mem.Allocator, kernel: anytype, border: BorderMode) !void {
            _ = allocator;
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            const kh = kernel.len;
            const kw = kernel[0].len;
            var flat: [kh * kw]f32 = undefined;
            inline for (0..kh) |r| inline for (0..kw) |cc| {
                flat[r * kw + cc] = zignal.meta.as(f32, kernel[r][cc]);
            };
            try check(c.zg_convolve_host(&desc(self.base), &desc(out.base), &flat, kh, kw, @intFromEnum(border)));
        }

        /// reference src/image.zig:635-648
        pub fn boxBlur(self: Self, out: Self, allocator: std.mem.Allocator, radius: u32) !void {
            _ = allocator;
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            try check(c.zg_box_blur_host(&desc(self.base), &desc(out.base), radius));
        }

        /// reference src/image.zig:653-783: medianBlur / percentileBlur / minBlur / maxBlur share op 0, midpointBlur is op 1,
        /// alphaTrimmedMeanBlur op 2 (InvalidPercentile / InvalidTrim are decided here so callers keep the reference's errors)
        pub fn percentileBlur(self: Self, out: Self, allocator: std.mem.Allocator, radius: usize, percentile: f64, border: BorderMode) !void {
            _ = allocator;
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            if (radius != 0 and (percentile < 0.0 or percentile > 1.0)) return error.InvalidPercentile;
            try check(c.zg_order_statistic_blur_host(&desc(self.base), &desc(out.base), @intCast(radius), 0, percentile, @intFromEnum(border)));
        }
        pub fn medianBlur(self: Self, out: Self, allocator: std.mem.Allocator, radius: usize) !void {
            return self.percentileBlur(out, allocator, radius, 0.5, .mirror);
        }
        pub fn midpointBlur(self: Self, out: Self, allocator: std.mem.Allocator, radius: usize, border: BorderMode) !void {
            _ = allocator;
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            try check(c.zg_order_statistic_blur_host(&desc(self.base), &desc(out.base), @intCast(radius), 1, 0.0, @intFromEnum(border)));
        }
        pub fn alphaTrimmedMeanBlur(self: Self, out: Self, allocator: std.mem.Allocator, radius: usize, trim_fraction: f64, border: BorderMode) !void {
            _ = allocator;
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            if (!std.math.isFinite(trim_fraction) or trim_fraction < 0.0 or trim_fraction >= 0.5) return error.InvalidTrim;
            try check(c.zg_order_statistic_blur_host(&desc(self.base), &desc(out.base), @intCast(radius), 2, trim_fraction, @intFromEnum(border)));
        }

        /// reference src/image.zig:804-829 (in place)
        pub fn autocontrast(self: Self, cutoff: f32) !void {
            if (cutoff < 0 or cutoff >= 0.5) return error.InvalidCutoff;
            try check(c.zg_autocontrast_host(&desc(self.base), cutoff));
        }
        pub fn equalize(self: Self) void {
            check(c.zg_equalize_host(&desc(self.base))) catch unreachable;
        }

        /// reference src/image.zig:845-914 (Image(u8) only, as there)
        pub fn thresholdOtsu(self: Self, out: Image(u8), allocator: std.mem.Allocator) !u8 {
            _ = allocator;
            if (comptime T != u8) @compileError("thresholdOtsu is only available for Image(u8)");
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            var t: u8 = 0;
            try check(c.zg_threshold_otsu_host(&desc(self.base), &Image(u8).desc(out.base), &t));
            return t;
        }
        pub fn thresholdAdaptiveMean(self: Self, out: Image(u8), allocator: std.mem.Allocator, radius: usize, cc: f32) !void {
            _ = allocator;
            if (comptime T != u8) @compileError("thresholdAdaptiveMean is only available for Image(u8)");
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            if (radius == 0) return error.InvalidRadius;
            try check(c.zg_threshold_adaptive_mean_host(&desc(self.base), &Image(u8).desc(out.base), @intCast(radius), cc));
        }
        fn morph(self: Self, out: Image(u8), kernel: zignal.BinaryKernel, iterations: usize, op: c_int) !void {
            if (comptime T != u8) @compileError("binary morphology is only available for Image(u8)");
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            try check(c.zg_morph_host(&desc(self.base), &Image(u8).desc(out.base), kernel.data.ptr, @intCast(kernel.rows), @intCast(kernel.cols), @intCast(iterations), op));
        }
        pub fn dilateBinary(self: Self, out: Image(u8), _: std.mem.Allocator, kernel: zignal.BinaryKernel, iterations: usize) !void { return self.morph(out, kernel, iterations, 0); }
        pub fn erodeBinary(self: Self, out: Image(u8), _: std.mem.Allocator, kernel: zignal.BinaryKernel, iterations: usize) !void { return self.morph(out, kernel, iterations, 1); }
        pub fn openBinary(self: Self, out: Image(u8), _: std.mem.Allocator, kernel: zignal.BinaryKernel, iterations: usize) !void { return self.morph(out, kernel, iterations, 2); }
        pub fn closeBinary(self: Self, out: Image(u8), _: std.mem.Allocator, kernel: zignal.BinaryKernel, iterations: usize) !void { return self.morph(out, kernel, iterations, 3); }

        /// reference src/image.zig:785-801
        pub fn sharpen(self: Self, out: Self, allocator: std.mem.Allocator, radius: usize) !void {
            _ = allocator;
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            try check(c.zg_sharpen_host(&desc(self.base), &desc(out.base), @intCast(radius)));
        }

        /// reference src/image.zig:494-513 (in place; Image(f32) is a compile error there and stays one here)
        pub fn invert(self: Self) void {
            if (T == f32) @compileError("invert() requires pixel types with an invert() method or u8 grayscale pixels");
            check(c.zg_invert_host(&desc(self.base))) catch unreachable;
        }

        /// reference src/image.zig:1001-1010 (fused grey -> Sobel x/y -> magnitude on the device)
        pub fn sobel(self: Self, out: Image(u8), allocator: std.mem.Allocator) !void {
            _ = allocator;
            if (self.base.rows != out.base.rows or self.base.cols != out.base.cols) return error.DimensionMismatch;
            try check(c.zg_sobel_host(&desc(self.base), &Image(u8).desc(out.base)));
        }

        /// reference src/image.zig:1015-1027; `opts.validate()` runs here so the reference's four distinct errors survive
        pub fn shenCastan(self: Self, out: Image(u8), allocator: std.mem.Allocator, opts: zignal.ShenCastan) !void {
            _ = allocator;
            if (self.base.rows != out.base.rows or self.base.cols != out.base.cols) return error.DimensionMismatch;
            try opts.validate();
            try check(c.zg_shen_castan_host(&desc(self.base), &Image(u8).desc(out.base), opts.smooth, @intCast(opts.window_size), opts.high_ratio, opts.low_rel,
                @intFromBool(opts.hysteresis), @intFromBool(opts.use_nms)));
        }

        /// reference src/image.zig:1077-1091 (MotionBlur is the reference's own union, src/image/motion_blur.zig:12-56)
        pub fn motionBlur(self: Self, out: Self, allocator: std.mem.Allocator, motion: MotionBlur) !void {
            _ = allocator;
            if (!self.base.hasSameShape(out.base)) return error.DimensionMismatch;
            switch (motion) {
                .linear => |p| try check(c.zg_motion_blur_linear_host(&desc(self.base), &desc(out.base), p.angle, @cos(p.angle), @sin(p.angle), @intCast(p.distance))),
                .radial_zoom => |p| try check(c.zg_motion_blur_radial_host(&desc(self.base), &desc(out.base), p.center_x, p.center_y, p.strength, 0)),
                .radial_spin => |p| try check(c.zg_motion_blur_radial_host(&desc(self.base), &desc(out.base), p.center_x, p.center_y, p.strength, 1)),
            }
        }

        /// reference src/image.zig:1047-1063. The reference's distinct errors are decided here, before the device call,
        /// so callers keep matching on error.InvalidParameter / InvalidSigma / InvalidThreshold (edges.zig:221-227).
        pub fn canny(self: Self, out: Image(u8), allocator: std.mem.Allocator, sigma: f32, low_threshold: f32, high_threshold: f32) !void {
            _ = allocator;
            if (self.base.rows != out.base.rows or self.base.cols != out.base.cols) return error.DimensionMismatch;
            if (!std.math.isFinite(sigma) or !std.math.isFinite(low_threshold) or !std.math.isFinite(high_threshold)) return error.InvalidParameter;
            if (sigma < 0) return error.InvalidSigma;
            if (low_threshold < 0 or high_threshold < 0 or low_threshold >= high_threshold) return error.InvalidThreshold;
            try check(c.zg_canny_host(&desc(self.base), &Image(u8).desc(out.base), sigma, low_threshold, high_threshold));
        }

        /// reference src/image.zig:523-525 (void: never fails; a HIP failure is a programming error here)
        pub fn resize(self: Self, out: Self, allocator: std.mem.Allocator, method: Interpolation) void {
            if (comptime isRgbU8(T)) {
                if (method == .lanczos and (self.base.rows != out.base.rows or self.base.cols != out.base.cols) and self.base.rows > 0 and self.base.cols > 0 and out.base.rows > 0 and out.base.cols > 0) {
                    // the plane kernel's weights come from @sin (channel_ops.zig:446-454): made here, in Zig
                    if (lanczosPlaneWeights(allocator, self.base.cols, out.base.cols)) |wx| {
                        defer allocator.free(wx);
                        if (lanczosPlaneWeights(allocator, self.base.rows, out.base.rows)) |wy| {
                            defer allocator.free(wy);
                            check(c.zg_resize_lanczos_weights_host(&desc(self.base), &desc(out.base), wx.ptr, wy.ptr)) catch unreachable;
                            return;
                        } else |_| {}
                    } else |_| {}
                }
            }
            check(c.zg_resize_host(&desc(self.base), &desc(out.base), &methodOf(method, null))) catch unreachable;
        }

        /// reference src/image.zig:530-541
        pub fn scale(self: Self, allocator: std.mem.Allocator, factor: f32, method: Interpolation) !Self {
            if (factor <= 0) return error.InvalidScaleFactor;
            const new_rows: u32 = @round(@as(f32, @floatFromInt(self.base.rows)) * factor);
            const new_cols: u32 = @round(@as(f32, @floatFromInt(self.base.cols)) * factor);
            if (new_rows == 0 or new_cols == 0) return error.InvalidDimensions;
            const scaled: Self = .{ .base = try .init(allocator, new_rows, new_cols) };
            self.resize(scaled, allocator, method);
            return scaled;
        }

        /// reference src/image.zig:546-548
        pub fn letterbox(self: Self, out: Self, allocator: std.mem.Allocator, method: Interpolation) Rectangle(u32) {
            _ = allocator;
            var r: [4]u32 = undefined;
            check(c.zg_letterbox_host(&desc(self.base), &desc(out.base), &methodOf(method, null), &r)) catch unreachable;
            return .init(r[0], r[1], r[2], r[3]);
        }

        /// reference src/image.zig:621-623 — `transform` is a Similarity / Affine / ProjectiveTransform(f32).
        pub fn warp(self: Self, out: Self, transform: anytype, method: Interpolation) void {
            const Tr = @TypeOf(transform);
            if (@hasField(Tr, "bias")) {
                const m = [6]f32{ transform.matrix.items[0][0], transform.matrix.items[0][1], transform.matrix.items[1][0], transform.matrix.items[1][1], transform.bias.items[0][0], transform.bias.items[1][0] };
                check(c.zg_warp_host(&desc(self.base), &desc(out.base), 1, &m, &methodOf(method, null))) catch unreachable;
            } else {
                var m: [9]f32 = undefined;
                inline for (0..3) |r| inline for (0..3) |cc| {
                    m[r * 3 + cc] = transform.matrix.items[r][cc];
                };
                check(c.zg_warp_host(&desc(self.base), &desc(out.base), 2, &m, &methodOf(method, null))) catch unreachable;
            }
        }

        /// reference src/image.zig:566-568 — @cos / @sin evaluated in Zig.
        pub fn rotateInto(self: Self, out: Self, angle: f32, method: Interpolation, border: BorderMode) void {
            check(c.zg_rotate_into_host(&desc(self.base), &desc(out.base), angle, @cos(angle), @sin(angle), &methodOf(method, null), @intFromEnum(border))) catch unreachable;
        }

        /// reference src/image.zig:558-562
        pub fn rotate(self: Self, allocator: std.mem.Allocator, angle: f32, method: Interpolation, border: BorderMode) !Self {
            const bounds = self.base.rotateBounds(angle); // pure host arithmetic, stays zignal's
            const rotated: Self = .{ .base = try .init(allocator, bounds.rows, bounds.cols) };
            self.rotateInto(rotated, angle, method, border);
            return rotated;
        }

        /// reference src/image.zig:593-595
        pub fn extract(self: Self, out: Self, rect: Rectangle(f32), angle: f32, method: Interpolation, border: BorderMode) void {
            const r = [4]f32{ rect.l, rect.t, rect.r, rect.b };
            check(c.zg_extract_host(&desc(self.base), &desc(out.base), &r, angle, @cos(angle), @sin(angle), &methodOf(method, null), @intFromEnum(border))) catch unreachable;
        }

        /// reference src/image.zig:582-584
        pub fn crop(self: Self, allocator: std.mem.Allocator, rectangle: Rectangle(f32)) !Self {
            const chip_rows: u32 = @round(rectangle.height());
            const chip_cols: u32 = @round(rectangle.width());
            const chip: Self = .{ .base = try .init(allocator, chip_rows, chip_cols) };
            self.extract(chip, rectangle, 0, .nearest, .zero);
            return chip;
        }

        /// reference src/image.zig:606-608 (all thirteen Blending modes on the device)
        pub fn insert(self: *Self, source: anytype, rect: Rectangle(f32), angle: f32, method: Interpolation, blend_mode: zignal.Blending) void {
            const r = [4]f32{ rect.l, rect.t, rect.r, rect.b };
            check(c.zg_insert_host(&desc(self.base), &desc(source.base), &r, angle, @cos(angle), @sin(angle), &methodOf(method, null), @intFromEnum(blend_mode))) catch unreachable;
        }

        /// reference src/image.zig:187-196
        pub fn fill(self: Self, value: T) void {
            check(c.zg_fill_host(&desc(self.base), &value)) catch unreachable;
        }
        /// reference src/image.zig:200-227 (rect is clipped to the image; no overlap fills everything)
        pub fn setBorder(self: Self, rect: Rectangle(u32), value: T) void {
            const r = [4]u32{ rect.l, rect.t, rect.r, rect.b };
            check(c.zg_set_border_host(&desc(self.base), &r, &value)) catch unreachable;
        }

        /// reference src/image/transforms.zig:28-44
        pub fn flipLeftRight(self: Self) void {
            check(c.zg_flip_left_right_host(&desc(self.base))) catch unreachable;
        }
        pub fn flipTopBottom(self: Self) void {
            check(c.zg_flip_top_bottom_host(&desc(self.base))) catch unreachable;
        }

        /// reference src/image.zig:396-407 — colour spaces on the GPU path: gray, rgb, rgba, oklab, xyz, ycbcr.
        pub fn convertInto(self: Self, comptime Target: type, out: Image(Target)) void {
            const lut = srgbLut();
            const src_space: c_int = switch (pixelOf(T)) { .u8, .f32 => 0, .rgb_u8, .rgb_f32 => 1, .rgba_u8, .rgba_f32 => 2 };
            const dst_space: c_int = comptime if (Target == u8 or Target == f32) 0 else switch (Target.space) {
                .gray => 0, .rgb => 1, .rgba => 2, .oklab => 3, .xyz => 4, .ycbcr => 5,
                else => @compileError("colour space not on the GPU hot path"),
            };
            check(c.zg_convert_host(&desc(self.base), src_space, &Image(Target).desc(out.base), dst_space, &lut)) catch unreachable;
        }

        /// reference src/image.zig:418-422
        pub fn convert(self: Self, allocator: std.mem.Allocator, comptime Target: type) !Image(Target) {
            const result: Image(Target) = .{ .base = try .init(allocator, self.base.rows, self.base.cols) };
            self.convertInto(Target, result);
            return result;
        }
    };
}

/// `Image(T)` whose pixels live in HBM: the device-resident face of the drop-in. The reference's `Image(T)` owns host memory
/// from a std.mem.Allocator (src/image.zig:124-134, deinit :173); this one owns a zg_malloc block and a stream to order its
/// work on. Methods have the reference's names and argument meaning but call the stream-taking entry points (zg_<op>, no
/// `_host`): they return as soon as the work is enqueued, and a chain such as the CLI's `pipeline [blur, resize]`
/// (src/cli/pipeline.zig:153-179) keeps every intermediate image on the GPU — PCIe is crossed once on the way in
/// (`upload` / `fromHost`) and once on the way out (`download` / `toHost`). Zig's own maths still makes every number
/// that depends on it (Gaussian taps with @exp, @cos / @sin of rotation angles, the sRGB table with std.math.pow).
/// Allocator parameters are kept where the reference has them so call sites do not change; they allocate host scratch only
/// (tap arrays), never pixels.
/// ImagePyramid(T) (reference src/image/pyramid.zig:11-170) resident on the device. `build` has the reference's signature and rules:
/// level 0 is the source itself (not copied, pyramid.zig:54), level i the source blurred with sigma_i = blur_sigma * sqrt(scale_i^2 - 1)
/// (only above 0.5) and resized bilinearly to trunc(dim / scale_i); a level below 8 x 8 truncates the pyramid. scale_i is computed HERE
/// with std.math.pow, so a Zig host keeps Zig's own bits; the whole pyramid is then one zg_pyramid_build call (no host round trip
/// between levels; the levels fork over internal streams under capture and join back into the source's stream).
pub fn DevicePyramid(comptime T: type) type {
    return struct {
        const Self = @This();
        levels: []DeviceImage(T),
        scale_factor: f32,
        n_levels: u8,
        blur_sigma: f32,
        allocator: std.mem.Allocator,

        pub fn build(allocator: std.mem.Allocator, source: DeviceImage(T), n_levels: u8, scale_factor: f32, blur_sigma: f32) !Self {
            std.debug.assert(n_levels > 0 and scale_factor > 1.0 and blur_sigma > 0);
            var levels = try allocator.alloc(DeviceImage(T), n_levels);
            errdefer allocator.free(levels);
            levels[0] = source;
            levels[0].owned = false;
            var descs = try allocator.alloc(c.ZgImage, n_levels);
            defer allocator.free(descs);
            var sigmas = try allocator.alloc(f32, n_levels);
            defer allocator.free(sigmas);
            var built: usize = 1;
            errdefer for (levels[1..built]) |*l| l.deinit();
            while (built < n_levels) : (built += 1) {
                const scale = std.math.pow(f32, scale_factor, @as(f32, @floatFromInt(built)));
                var r: u32 = 0;
                var cc: u32 = 0;
                var sigma: f32 = 0;
                try check(c.zg_pyramid_level(source.rows, source.cols, scale, blur_sigma, &r, &cc, &sigma));
                if (r < 8 or cc < 8) break; // pyramid.zig:63-73
                levels[built] = try DeviceImage(T).init(r, cc, source.stream);
                descs[built - 1] = levels[built].desc();
                sigmas[built - 1] = sigma;
            }
            const s = source.desc();
            try check(c.zg_pyramid_build(&s, descs.ptr, sigmas.ptr, @intCast(built - 1), source.stream));
            return .{ .levels = try allocator.realloc(levels, built), .scale_factor = scale_factor, .n_levels = @intCast(built), .blur_sigma = blur_sigma, .allocator = allocator };
        }
        pub fn buildDefault(allocator: std.mem.Allocator, source: DeviceImage(T)) !Self { // pyramid.zig:105-107
            return build(allocator, source, 8, 1.2, 1.6);
        }
        pub fn deinit(self: *Self) void { // pyramid.zig:110-118: level 0 belongs to the caller
            for (self.levels[1..]) |*l| l.deinit();
            self.allocator.free(self.levels);
        }
        pub fn getScale(self: Self, level: usize) f32 { // pyramid.zig:122-125
            return std.math.pow(f32, self.scale_factor, @as(f32, @floatFromInt(level)));
        }
    };
}

pub fn DeviceImage(comptime T: type) type {
    return struct {
        const Self = @This();
        rows: u32 = 0,
        cols: u32 = 0,
        stride: usize = 0,
        data: ?*anyopaque = null, // device pointer: never dereferenced on the host
        stream: ?*anyopaque = null, // zg_stream; null = the default stream
        owned: bool = false,

        fn desc(self: Self) c.ZgImage {
            return .{ .data = self.data, .stride = self.stride, .rows = self.rows, .cols = self.cols, .pixel = @intFromEnum(pixelOf(T)) };
        }
        fn hostDesc(img: zignal.Image(T)) c.ZgImage {
            return .{ .data = @ptrCast(img.data.ptr), .stride = img.stride, .rows = img.rows, .cols = img.cols, .pixel = @intFromEnum(pixelOf(T)) };
        }

        /// reference src/image.zig:124-134, in HBM
        pub fn init(rows: u32, cols: u32, stream: ?*anyopaque) !Self {
            var p: ?*anyopaque = null;
            try check(c.zg_malloc(&p, @as(usize, rows) * cols * @sizeOf(T)));
            return .{ .rows = rows, .cols = cols, .stride = cols, .data = p, .stream = stream, .owned = true };
        }
        /// reference src/image.zig:173
        pub fn deinit(self: *Self) void {
            if (self.owned) _ = c.zg_free(self.data); // waits for work that still uses the block
            self.* = .{};
        }
        pub fn fromHost(host: zignal.Image(T), stream: ?*anyopaque) !Self {
            var img = try init(host.rows, host.cols, stream);
            errdefer img.deinit();
            try img.upload(host);
            return img;
        }
        /// one trip across PCIe, strides honoured on both sides; complete on return
        pub fn upload(self: Self, host: zignal.Image(T)) !void {
            try check(c.zg_image_upload(&self.desc(), &hostDesc(host), self.stream));
        }
        /// waits for the stream's work on this image, then one trip back
        pub fn download(self: Self, host: zignal.Image(T)) !void {
            try check(c.zg_image_download(&hostDesc(host), &self.desc(), self.stream));
        }
        pub fn toHost(self: Self, allocator: std.mem.Allocator) !zignal.Image(T) {
            const host: zignal.Image(T) = try .init(allocator, self.rows, self.cols);
            errdefer host.deinit(allocator);
            try self.download(host);
            return host;
        }
        pub fn synchronize(self: Self) !void {
            try check(c.zg_stream_synchronize(self.stream));
        }
        /// reference src/image.zig:332-352 (non-owning)
        pub fn view(self: Self, rect: Rectangle(u32)) Self {
            const l = rect.l;
            const t = rect.t;
            const r = @min(rect.r, self.cols);
            const b = @min(rect.b, self.rows);
            if (l >= r or t >= b) return .{ .stream = self.stream };
            const offset = (@as(usize, t) * self.stride + l) * @sizeOf(T);
            return .{ .rows = b - t, .cols = r - l, .stride = self.stride, .data = @ptrFromInt(@intFromPtr(self.data.?) + offset), .stream = self.stream, .owned = false };
        }
        pub fn hasSameShape(self: Self, other: anytype) bool {
            return self.rows == other.rows and self.cols == other.cols;
        }

        /// reference src/image.zig:935-951
        pub fn convolveSeparable(self: Self, out: Self, allocator: std.mem.Allocator, kernel_x: []const f32, kernel_y: []const f32, border: BorderMode) !void {
            _ = allocator;
            if (!self.hasSameShape(out)) return error.DimensionMismatch;
            try check(c.zg_conv_separable(&self.desc(), &out.desc(), kernel_x.ptr, @intCast(kernel_x.len), kernel_y.ptr, @intCast(kernel_y.len), @intFromEnum(border), self.stream));
        }
        /// reference src/image.zig:954-994 — taps built here with Zig's @exp (the library copies them before returning)
        pub fn gaussianBlur(self: Self, out: Self, allocator: std.mem.Allocator, sigma: f32) !void {
            if (!self.hasSameShape(out)) return error.DimensionMismatch;
            if (sigma == 0) return check(c.zg_copy(&self.desc(), &out.desc(), self.stream));
            if (sigma < 0) return error.InvalidSigma;
            const radius: usize = @ceil(3.0 * sigma);
            const kernel = try allocator.alloc(f32, 2 * radius + 1);
            defer allocator.free(kernel);
            var sum: f32 = 0;
            for (kernel, 0..) |*k, i| {
                const x = @as(f32, @floatFromInt(i)) - @as(f32, @floatFromInt(radius));
                k.* = @exp(-(x * x) / (2.0 * sigma * sigma));
                sum += k.*;
            }
            for (kernel) |*k| k.* /= sum;
            try self.convolveSeparable(out, allocator, kernel, kernel, .mirror);
        }
        /// convolveSeparable over several planes of one shape in one launch (zg_conv_separable_planes). The reference's f32 route
        /// is per plane — Image(Rgba(f32)).convolveSeparable is a compile error (src/image/convolution.zig:431-435) — so a host that
        /// keeps RGBA f32 data as four Image(f32) planes calls this with all four instead of paying four launches. Asynchronous on
        /// planes[0].stream; every pair is checked like the single-plane call.
        pub fn convolveSeparablePlanes(planes: []const Self, outs: []const Self, allocator: std.mem.Allocator, kernel_x: []const f32, kernel_y: []const f32, border: BorderMode) !void {
            if (planes.len != outs.len) return error.DimensionMismatch;
            if (planes.len == 0) return;
            const descs = try allocator.alloc(c.ZgImage, 2 * planes.len);
            defer allocator.free(descs);
            for (planes, outs, 0..) |p, o, i| {
                if (!p.hasSameShape(o)) return error.DimensionMismatch;
                descs[i] = p.desc();
                descs[planes.len + i] = o.desc();
            }
            try check(c.zg_conv_separable_planes(descs.ptr, descs.ptr + planes.len, @intCast(planes.len), kernel_x.ptr, @intCast(kernel_x.len), kernel_y.ptr, @intCast(kernel_y.len), @intFromEnum(border), planes[0].stream));
        }
        /// gaussianBlur (reference src/image.zig:954-994) over several planes in one launch; the taps are Zig's own (@exp), as in gaussianBlur.
        pub fn gaussianBlurPlanes(planes: []const Self, outs: []const Self, allocator: std.mem.Allocator, sigma: f32) !void {
            if (planes.len != outs.len) return error.DimensionMismatch;
            if (sigma < 0) return error.InvalidSigma;
            if (sigma == 0) {
                for (planes, outs) |p, o| try p.gaussianBlur(o, allocator, 0);
                return;
            }
            const radius: usize = @ceil(3.0 * sigma);
            const kernel = try allocator.alloc(f32, 2 * radius + 1);
            defer allocator.free(kernel);
            var sum: f32 = 0;
            for (kernel, 0..) |*k, i| {
                const x = @as(f32, @floatFromInt(i)) - @as(f32, @floatFromInt(radius));
                k.* = @exp(-(x * x) / (2.0 * sigma * sigma));
                sum += k.*;
            }
            for (kernel) |*k| k.* /= sum;
            try convolveSeparablePlanes(planes, outs, allocator, kernel, kernel, .mirror);
        }
        /// reference src/image.zig:917-932
        pub fn convolve(self: Self, out: Self, allocator: std.mem.Allocator, kernel: anytype, border: BorderMode) !void {
            _ = allocator;
            if (!self.hasSameShape(out)) return error.DimensionMismatch;
            const kh = kernel.len;
            const kw = kernel[0].len;
            var flat: [kh * kw]f32 = undefined;
            inline for (0..kh) |r| inline for (0..kw) |cc| {
                flat[r * kw + cc] = zignal.meta.as(f32, kernel[r][cc]);
            };
            try check(c.zg_convolve(&self.desc(), &out.desc(), &flat, kh, kw, @intFromEnum(border), self.stream));
        }
        /// reference src/image.zig:635-648
        pub fn boxBlur(self: Self, out: Self, allocator: std.mem.Allocator, radius: u32) !void {
            _ = allocator;
            if (!self.hasSameShape(out)) return error.DimensionMismatch;
            try check(c.zg_box_blur(&self.desc(), &out.desc(), radius, self.stream));
        }
        /// reference src/image.zig:1001-1010
        pub fn sobel(self: Self, out: DeviceImage(u8), allocator: std.mem.Allocator) !void {
            _ = allocator;
            if (!self.hasSameShape(out)) return error.DimensionMismatch;
            try check(c.zg_sobel(&self.desc(), &out.desc(), self.stream));
        }
        /// reference src/image.zig:1047-1063
        pub fn canny(self: Self, out: DeviceImage(u8), allocator: std.mem.Allocator, sigma: f32, low_threshold: f32, high_threshold: f32) !void {
            _ = allocator;
            if (!self.hasSameShape(out)) return error.DimensionMismatch;
            if (!std.math.isFinite(sigma) or !std.math.isFinite(low_threshold) or !std.math.isFinite(high_threshold)) return error.InvalidParameter;
            if (sigma < 0) return error.InvalidSigma;
            if (low_threshold < 0 or high_threshold < 0 or low_threshold >= high_threshold) return error.InvalidThreshold;
            try check(c.zg_canny(&self.desc(), &out.desc(), sigma, low_threshold, high_threshold, self.stream));
        }
        /// reference src/image.zig:523-525
        pub fn resize(self: Self, out: Self, allocator: std.mem.Allocator, method: Interpolation) void {
            if (comptime isRgbU8(T)) {
                if (method == .lanczos and (self.rows != out.rows or self.cols != out.cols) and self.rows > 0 and self.cols > 0 and out.rows > 0 and out.cols > 0) {
                    if (lanczosPlaneWeights(allocator, self.cols, out.cols)) |wx| {
                        defer allocator.free(wx);
                        if (lanczosPlaneWeights(allocator, self.rows, out.rows)) |wy| {
                            defer allocator.free(wy);
                            check(c.zg_resize_lanczos_weights(&self.desc(), &out.desc(), wx.ptr, wy.ptr, self.stream)) catch unreachable;
                            return;
                        } else |_| {}
                    } else |_| {}
                }
            }
            check(c.zg_resize(&self.desc(), &out.desc(), &methodOf(method, null), self.stream)) catch unreachable;
        }
        /// reference src/image.zig:530-541
        pub fn scale(self: Self, allocator: std.mem.Allocator, factor: f32, method: Interpolation) !Self {
            if (factor <= 0) return error.InvalidScaleFactor;
            const new_rows: u32 = @round(@as(f32, @floatFromInt(self.rows)) * factor);
            const new_cols: u32 = @round(@as(f32, @floatFromInt(self.cols)) * factor);
            if (new_rows == 0 or new_cols == 0) return error.InvalidDimensions;
            const scaled = try init(new_rows, new_cols, self.stream);
            self.resize(scaled, allocator, method);
            return scaled;
        }
        /// reference src/image.zig:546-548
        pub fn letterbox(self: Self, out: Self, allocator: std.mem.Allocator, method: Interpolation) Rectangle(u32) {
            _ = allocator;
            var r: [4]u32 = undefined;
            check(c.zg_letterbox(&self.desc(), &out.desc(), &methodOf(method, null), &r, self.stream)) catch unreachable;
            return .init(r[0], r[1], r[2], r[3]);
        }
        /// reference src/image.zig:621-623
        pub fn warp(self: Self, out: Self, transform: anytype, method: Interpolation) void {
            const Tr = @TypeOf(transform);
            if (@hasField(Tr, "bias")) {
                const m = [6]f32{ transform.matrix.items[0][0], transform.matrix.items[0][1], transform.matrix.items[1][0], transform.matrix.items[1][1], transform.bias.items[0][0], transform.bias.items[1][0] };
                check(c.zg_warp(&self.desc(), &out.desc(), 1, &m, &methodOf(method, null), self.stream)) catch unreachable;
            } else {
                var m: [9]f32 = undefined;
                inline for (0..3) |r| inline for (0..3) |cc| {
                    m[r * 3 + cc] = transform.matrix.items[r][cc];
                };
                check(c.zg_warp(&self.desc(), &out.desc(), 2, &m, &methodOf(method, null), self.stream)) catch unreachable;
            }
        }
        /// reference src/image.zig:566-568 — @cos / @sin evaluated in Zig
        pub fn rotateInto(self: Self, out: Self, angle: f32, method: Interpolation, border: BorderMode) void {
            check(c.zg_rotate_into(&self.desc(), &out.desc(), angle, @cos(angle), @sin(angle), &methodOf(method, null), @intFromEnum(border), self.stream)) catch unreachable;
        }
        /// reference src/image.zig:593-595
        pub fn extract(self: Self, out: Self, rect: Rectangle(f32), angle: f32, method: Interpolation, border: BorderMode) void {
            const r = [4]f32{ rect.l, rect.t, rect.r, rect.b };
            check(c.zg_extract(&self.desc(), &out.desc(), &r, angle, @cos(angle), @sin(angle), &methodOf(method, null), @intFromEnum(border), self.stream)) catch unreachable;
        }
        /// reference src/image.zig:582-584
        pub fn crop(self: Self, allocator: std.mem.Allocator, rectangle: Rectangle(f32)) !Self {
            _ = allocator;
            const chip_rows: u32 = @round(rectangle.height());
            const chip_cols: u32 = @round(rectangle.width());
            const chip = try init(chip_rows, chip_cols, self.stream);
            self.extract(chip, rectangle, 0, .nearest, .zero);
            return chip;
        }
        /// reference src/image.zig:187-196
        pub fn fill(self: Self, value: T) void {
            check(c.zg_fill(&self.desc(), &value, self.stream)) catch unreachable;
        }
        /// reference src/image/transforms.zig:28-44
        pub fn flipLeftRight(self: Self) void {
            check(c.zg_flip_left_right(&self.desc(), self.stream)) catch unreachable;
        }
        pub fn flipTopBottom(self: Self) void {
            check(c.zg_flip_top_bottom(&self.desc(), self.stream)) catch unreachable;
        }
        /// reference src/image.zig:396-407
        pub fn convertInto(self: Self, comptime Target: type, out: DeviceImage(Target)) void {
            const lut = srgbLut();
            const src_space: c_int = switch (pixelOf(T)) { .u8, .f32 => 0, .rgb_u8, .rgb_f32 => 1, .rgba_u8, .rgba_f32 => 2 };
            const dst_space: c_int = comptime if (Target == u8 or Target == f32) 0 else switch (Target.space) {
                .gray => 0, .rgb => 1, .rgba => 2, .oklab => 3, .xyz => 4, .ycbcr => 5,
                else => @compileError("colour space not on the GPU hot path"),
            };
            check(c.zg_convert(&self.desc(), src_space, &out.desc(), dst_space, &lut, self.stream)) catch unreachable;
        }
        /// resize(out-sized, method) followed by convertInto(Target) as one call (zg_resize_convert): the pipeline steps
        /// [resize, convert] of src/cli/pipeline.zig:153-179 without the intermediate image.
        pub fn resizeConvertInto(self: Self, comptime Target: type, out: DeviceImage(Target), method: Interpolation) void {
            const lut = srgbLut();
            const src_space: c_int = switch (pixelOf(T)) { .u8, .f32 => 0, .rgb_u8, .rgb_f32 => 1, .rgba_u8, .rgba_f32 => 2 };
            const dst_space: c_int = comptime if (Target == u8 or Target == f32) 0 else switch (Target.space) {
                .gray => 0, .rgb => 1, .rgba => 2, .oklab => 3, .xyz => 4, .ycbcr => 5,
                else => @compileError("colour space not on the GPU hot path"),
            };
            check(c.zg_resize_convert(&self.desc(), src_space, &out.desc(), dst_space, &methodOf(method, null), &lut, self.stream)) catch unreachable;
        }
        /// reference src/image.zig:418-422
        pub fn convert(self: Self, allocator: std.mem.Allocator, comptime Target: type) !DeviceImage(Target) {
            _ = allocator;
            const result = try DeviceImage(Target).init(self.rows, self.cols, self.stream);
            self.convertInto(Target, result);
            return result;
        }
    };
}


/// PNG through the library (reference src/codecs/png.zig): the chunk layer, inflate / deflate and de-filtering run on the
/// host inside libzignal_hip.so, unpacking / conversion / row filtering on the MI355X. The library reports the reference's
/// error names as text (zg_last_error() starts with the name); `codecError` turns the common ones back into the error set
/// a caller of png.zig already matches on, and keeps the rest as error.PngError.
pub const png = struct {
    pub const DecodeLimits = c.ZgPngLimits;
    pub const Header = c.ZgPngHeader;

    pub fn defaultLimits() DecodeLimits {
        var l: DecodeLimits = undefined;
        c.zg_png_default_limits(&l);
        return l;
    }

    fn codecError() anyerror {
        const msg = std.mem.span(c.zg_last_error());
        const name = msg[0 .. std.mem.indexOfScalar(u8, msg, ' ') orelse msg.len];
        const known = .{
            .{ "InvalidPngSignature", error.InvalidPngSignature }, .{ "InvalidCrc", error.InvalidCrc },
            .{ "ImageTooLarge", error.ImageTooLarge },             .{ "PngDataTooLarge", error.PngDataTooLarge },
            .{ "TooManyChunks", error.TooManyChunks },             .{ "MissingHeader", error.MissingHeader },
            .{ "MissingImageData", error.MissingImageData },       .{ "MissingPalette", error.MissingPalette },
            .{ "InvalidChunkLength", error.InvalidChunkLength },   .{ "ReadFailed", error.ReadFailed },
            .{ "InvalidFilterType", error.InvalidFilterType },     .{ "InvalidPaletteIndex", error.InvalidPaletteIndex },
            .{ "NonConsecutiveIdatChunks", error.NonConsecutiveIdatChunks },
        };
        inline for (known) |entry| if (std.mem.eql(u8, name, entry[0])) return entry[1];
        return error.PngError;
    }

    fn checkPng(status: c_int) !void {
        if (status == 6) return codecError();
        return check(status);
    }

    /// reference src/codecs/png.zig:308-410
    pub fn getInfo(data: []const u8, limits: DecodeLimits) !Header {
        var h: Header = undefined;
        try checkPng(c.zg_png_info(data.ptr, data.len, &limits, &h));
        return h;
    }

    /// reference src/codecs/png.zig:1151-1186
    pub fn loadFromBytes(comptime T: type, allocator: std.mem.Allocator, data: []const u8, limits: DecodeLimits) !Image(T) {
        var h: Header = undefined;
        try checkPng(c.zg_png_probe(data.ptr, data.len, &limits, &h, null, null));
        const out: Image(T) = .{ .base = try .init(allocator, h.height, h.width) };
        errdefer out.base.deinit(allocator);
        const space: c_int = switch (pixelOf(T)) { .u8, .f32 => 0, .rgb_u8, .rgb_f32 => 1, .rgba_u8, .rgba_f32 => 2 };
        try checkPng(c.zg_png_decode_host(data.ptr, data.len, &limits, &Image(T).desc(out.base), space, null));
        return out;
    }

    /// reference src/codecs/png.zig:1400-1425; the bytes are copied into `allocator`'s memory
    pub fn encode(comptime T: type, allocator: std.mem.Allocator, image: Image(T), options: ?*const c.ZgPngEncodeOptions) ![]u8 {
        var mem: ?[*]u8 = null;
        var len: usize = 0;
        const space: c_int = switch (pixelOf(T)) { .u8, .f32 => 0, .rgb_u8, .rgb_f32 => 1, .rgba_u8, .rgba_f32 => 2 };
        try checkPng(c.zg_png_encode_host(&Image(T).desc(image.base), space, options, &mem, &len));
        defer c.zg_png_free(mem);
        return allocator.dupe(u8, mem.?[0..len]);
    }
};

/// JPEG decoding through the library (reference src/codecs/jpeg.zig): markers and Huffman decoding on the host inside
/// libzignal_hip.so, dequantisation / IDCT / chroma / colour on the MI355X. Error names travel as text, as for `png`.
pub const jpeg = struct {
    pub const DecodeLimits = c.ZgJpegLimits;
    pub const Header = c.ZgJpegHeader;

    pub fn defaultLimits() DecodeLimits {
        var l: DecodeLimits = undefined;
        c.zg_jpeg_default_limits(&l);
        return l;
    }

    fn codecError() anyerror {
        const msg = std.mem.span(c.zg_last_error());
        const name = msg[0 .. std.mem.indexOfScalar(u8, msg, ' ') orelse msg.len];
        const known = .{
            .{ "InvalidJpegFile", error.InvalidJpegFile },             .{ "InvalidMarker", error.InvalidMarker },
            .{ "InvalidSOF", error.InvalidSOF },                       .{ "DuplicateSOF", error.DuplicateSOF },
            .{ "InvalidSOS", error.InvalidSOS },                       .{ "NoScanData", error.NoScanData },
            .{ "InvalidHuffmanCode", error.InvalidHuffmanCode },       .{ "MissingHuffmanTable", error.MissingHuffmanTable },
            .{ "MissingQuantTable", error.MissingQuantTable },         .{ "ImageTooLarge", error.ImageTooLarge },
            .{ "UnsupportedSamplingFactor", error.UnsupportedSamplingFactor },
            .{ "UnsupportedComponentCount", error.UnsupportedComponentCount },
            .{ "BlockMemoryLimitExceeded", error.BlockMemoryLimitExceeded },
        };
        inline for (known) |entry| if (std.mem.eql(u8, name, entry[0])) return entry[1];
        return error.JpegError;
    }

    fn checkJpeg(status: c_int) !void {
        if (status == 6) return codecError();
        return check(status);
    }

    /// reference src/codecs/jpeg.zig:77-179
    pub fn getInfo(data: []const u8, limits: DecodeLimits) !Header {
        var h: Header = undefined;
        try checkJpeg(c.zg_jpeg_info(data.ptr, data.len, &limits, &h));
        return h;
    }

    /// reference src/codecs/jpeg.zig:2825-2851
    pub fn loadFromBytes(comptime T: type, allocator: std.mem.Allocator, data: []const u8, limits: DecodeLimits) !Image(T) {
        const h = try getInfo(data, limits);
        const out: Image(T) = .{ .base = try .init(allocator, h.height, h.width) };
        errdefer out.base.deinit(allocator);
        const space: c_int = switch (pixelOf(T)) { .u8, .f32 => 0, .rgb_u8, .rgb_f32 => 1, .rgba_u8, .rgba_f32 => 2 };
        try checkJpeg(c.zg_jpeg_decode_host(data.ptr, data.len, &limits, &Image(T).desc(out.base), space, null));
        return out;
    }

    /// reference src/codecs/jpeg.zig:307-329; the bytes are copied into `allocator`'s memory
    pub fn encode(comptime T: type, allocator: std.mem.Allocator, image: Image(T), options: ?*const c.ZgJpegEncodeOptions) ![]u8 {
        var mem: ?[*]u8 = null;
        var len: usize = 0;
        const space: c_int = switch (pixelOf(T)) { .u8, .f32 => 0, .rgb_u8, .rgb_f32 => 1, .rgba_u8, .rgba_f32 => 2 };
        try checkJpeg(c.zg_jpeg_encode_host(&Image(T).desc(image.base), space, options, &mem, &len));
        defer c.zg_jpeg_free(mem);
        return allocator.dupe(u8, mem.?[0..len]);
    }
};
