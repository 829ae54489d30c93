//! zig_golden.zig — golden vectors for the transcendental boundary of zignal's image hot path, made by REAL Zig.
//!
//! The MI355X library and its CPU oracle restate Zig's std maths (musl's expf / sinf / cosf / cbrtf, Go's Pow) from the published
//! algorithms, because no Zig toolchain exists where they are built (DESIGN.md §4: "parity unpinned against Zig at the last ulp").
//! This program is the other half of that sentence: run it once with the toolchain the reference pins
//! (build.zig.zon: minimum_zig_version 0.17.0-dev.1441+d5181a9c9) and the pytest beside it turns "unpinned" into "pinned":
//!
//!     zig run -O ReleaseFast tools/zig_golden.zig > tests/golden/zig_golden.json
//!     python -m pytest tests/test_zig_golden.py -q          # skipped while the file is absent
//!
//! (ReleaseFast is what the reference's own CI and examples build with; Debug must give the same bits — none of this is fast-math.)
//! Std only: every expression below is the reference's own, restated with its file:line, so the numbers are those zignal computes
//! with this compiler — the same @exp, @sin, @cos, std.math.pow and std.math.cbrt calls on the same operands in the same order.
//! Every f32 travels as the u32 of its bit pattern; sweeps carry their inputs, so the checker never has to re-derive them.
//!
//! Sections of the JSON object:
//!   gamma_to_linear         256 x gammaToLinear(f32, i / 255)                  src/color.zig:1252-1258 (+ :365-373 for the / 255)
//!   lanczos3_lut_comptime   1025 x lanczosKernel(i / (1024 / 3), 3) at comptime  src/image/interpolation.zig:245-267 (the table the library takes in zg_method.lanczos_lut)
//!   lanczos3_lut_runtime    the same expression at run time (compiler-rt's sinf; shows whether comptime and run-time @sin agree)
//!   gaussian_taps           gaussianBlur's normalised taps for a sigma sweep   src/image.zig:973-990
//!   lanczos_plane_weights   resizePlaneLanczosU8's six weights per destination column, 4096 -> 1500 and 640 -> 1000   src/image/channel_ops.zig:446-466
//!   oklab_17 / lab_17       Rgb(u8) on the 17^3 lattice {0, 16, ..., 240, 255}^3 -> Oklab(f32) / Lab(f32)   src/color.zig:1261-1272, 1289-1310, 1381-1400
//!   exp / sin / cos / cbrt / pow24 / pow_third / pow_inv24   [input, output] pairs over the argument ranges the path uses
const std = @import("std");
const builtin = @import("builtin");

fn bits(x: f32) u32 {
    return @bitCast(x);
}

// ---- src/color.zig:84-89, 74-81 -----------------------------------------------------------------------------------------
const srgb_gamma_threshold = 0.04045;
const srgb_gamma_offset = 0.055;
const srgb_gamma_scale = 1.055;
const srgb_linear_slope = 12.92;
const srgb_gamma_exponent = 2.4;
const d65_x = 95.047;
const d65_y = 100.000;
const d65_z = 108.883;
const lab_epsilon = 0.008856;
const lab_kappa_div_116 = 7.787;
const lab_delta = 16.0 / 116.0;
const pow = std.math.pow;

/// src/color.zig:1252-1258
fn gammaToLinear(comptime T: type, c: T) T {
    return if (c > srgb_gamma_threshold)
        pow(T, (c + srgb_gamma_offset) / srgb_gamma_scale, srgb_gamma_exponent)
    else
        c / srgb_linear_slope;
}

const Xyz = struct { x: f32, y: f32, z: f32 };

/// src/color.zig:365-373 (Rgb(u8).as(f32): @as(U, self.r) / 255) then :1261-1272
fn rgbToXyz(r8: u8, g8: u8, b8: u8) Xyz {
    const T = f32;
    const r = gammaToLinear(T, @as(T, @floatFromInt(r8)) / 255);
    const g = gammaToLinear(T, @as(T, @floatFromInt(g8)) / 255);
    const b = gammaToLinear(T, @as(T, @floatFromInt(b8)) / 255);
    return .{
        .x = (r * 0.4124 + g * 0.3576 + b * 0.1805) * 100,
        .y = (r * 0.2126 + g * 0.7152 + b * 0.0722) * 100,
        .z = (r * 0.0193 + g * 0.1192 + b * 0.9505) * 100,
    };
}

/// src/color.zig:1381-1400
fn xyzToOklab(xyz: Xyz) [3]f32 {
    const x = xyz.x / 100.0;
    const y = xyz.y / 100.0;
    const z = xyz.z / 100.0;
    const l_linear = 0.8189330101 * x + 0.3618667424 * y - 0.1288597137 * z;
    const m_linear = 0.0329845436 * x + 0.9293118715 * y + 0.0361456387 * z;
    const s_linear = 0.0482003018 * x + 0.2643662691 * y + 0.6338517070 * z;
    const l_dash = std.math.cbrt(l_linear);
    const m_dash = std.math.cbrt(m_linear);
    const s_dash = std.math.cbrt(s_linear);
    return .{
        0.2104542553 * l_dash + 0.7936177850 * m_dash - 0.0040720468 * s_dash,
        1.9779984951 * l_dash - 2.4285922050 * m_dash + 0.4505937099 * s_dash,
        0.0259040371 * l_dash + 0.7827717662 * m_dash - 0.8086757660 * s_dash,
    };
}

/// src/color.zig:1289-1291
fn labForward(comptime T: type, t: T) T {
    return if (t > lab_epsilon) pow(T, t, 1.0 / 3.0) else lab_kappa_div_116 * t + lab_delta;
}

/// src/color.zig:1294-1310
fn xyzToLab(xyz: Xyz) [3]f32 {
    const T = f32;
    const fx = labForward(T, xyz.x / d65_x);
    const fy = labForward(T, xyz.y / d65_y);
    const fz = labForward(T, xyz.z / d65_z);
    return .{ @max(0, 116.0 * fy - 16.0), 500.0 * (fx - fy), 200.0 * (fy - fz) };
}

/// src/image/interpolation.zig:245-252
fn lanczosKernel(x: f32, a: f32) f32 {
    if (x == 0) return 1;
    if (@abs(x) >= a) return 0;
    const pi_x = std.math.pi * x;
    const pi_x_over_a = pi_x / a;
    return (a * @sin(pi_x) * @sin(pi_x_over_a)) / (pi_x * pi_x);
}

/// src/image/interpolation.zig:255-267, verbatim: evaluated by the COMPILER
const lanczos3_lut: [1025]f32 = blk: {
    const size = 1024;
    const max_dist: f32 = 3.0;
    const step = size / max_dist;
    @setEvalBranchQuota(40000);
    var vals: [size + 1]f32 = undefined;
    for (0..1025) |i| {
        const x = @as(f32, @floatFromInt(i)) / step;
        vals[i] = lanczosKernel(x, 3.0);
    }
    break :blk vals;
};

/// src/image/channel_ops.zig:446-454
fn lanczosPlaneKernel(x: f32) f32 {
    if (x == 0) return 1.0;
    const a = 3.0;
    if (@abs(x) >= a) return 0.0;
    const pi_x = std.math.pi * x;
    return (a * @sin(pi_x) * @sin(pi_x / a)) / (pi_x * pi_x);
}

// A run-time value the optimiser cannot fold: the sweeps and tables below must be computed by the generated code, not by the compiler.
fn runtime(x: f32) f32 {
    var v = x;
    std.mem.doNotOptimizeAway(&v);
    return v;
}

const Lcg = struct {
    s: u32,
    fn next(self: *Lcg) u32 {
        self.s = self.s *% 1664525 +% 1013904223;
        return self.s;
    }
    /// uniform in [lo, hi): a 24-bit fraction, one multiply and one add in f32
    fn uniform(self: *Lcg, lo: f32, hi: f32) f32 {
        const u = @as(f32, @floatFromInt(self.next() >> 8)) * (1.0 / 16777216.0);
        return lo + u * (hi - lo);
    }
};

fn sweep(w: anytype, comptime name: []const u8, comptime f: fn (f32) f32, seed: u32, lo: f32, hi: f32, n: usize, last: bool) !void {
    var rng = Lcg{ .s = seed };
    try w.print("  \"{s}\": [", .{name});
    for (0..n) |i| {
        const x = runtime(rng.uniform(lo, hi));
        try w.print("{s}[{d},{d}]", .{ if (i == 0) "" else ",", bits(x), bits(f(x)) });
    }
    try w.print("]{s}\n", .{if (last) "" else ","});
}

fn fExp(x: f32) f32 {
    return @exp(x);
}
fn fSin(x: f32) f32 {
    return @sin(x);
}
fn fCos(x: f32) f32 {
    return @cos(x);
}
fn fCbrt(x: f32) f32 {
    return std.math.cbrt(x);
}
fn fPow24(x: f32) f32 {
    return pow(f32, x, srgb_gamma_exponent);
}
fn fPowThird(x: f32) f32 {
    return pow(f32, x, 1.0 / 3.0);
}
fn fPowInv24(x: f32) f32 {
    return pow(f32, x, 1.0 / srgb_gamma_exponent);
}

pub fn main(init: std.process.Init) !void {
    var buffer: [1 << 16]u8 = undefined;
    var stdout = std.Io.File.stdout().writer(init.io, &buffer);
    const w = &stdout.interface;

    try w.print("{{\n  \"zig_version\": \"{s}\",\n  \"optimize\": \"{s}\",\n", .{ builtin.zig_version_string, @tagName(builtin.mode) });

    try w.print("  \"gamma_to_linear\": [", .{});
    for (0..256) |i| {
        const c = runtime(@as(f32, @floatFromInt(i))) / 255;
        try w.print("{s}{d}", .{ if (i == 0) "" else ",", bits(gammaToLinear(f32, c)) });
    }
    try w.print("],\n  \"lanczos3_lut_comptime\": [", .{});
    for (lanczos3_lut, 0..) |v, i| try w.print("{s}{d}", .{ if (i == 0) "" else ",", bits(v) });
    try w.print("],\n  \"lanczos3_lut_runtime\": [", .{});
    {
        const step = runtime(1024.0) / runtime(3.0);
        for (0..1025) |i| {
            const x = runtime(@as(f32, @floatFromInt(i))) / step;
            try w.print("{s}{d}", .{ if (i == 0) "" else ",", bits(lanczosKernel(x, 3.0)) });
        }
    }

    // src/image.zig:973-990
    try w.print("],\n  \"gaussian_taps\": {{", .{});
    const sigmas = [_]f32{ 0.3, 0.5, 0.6, 0.75, 1.0, 1.2, 1.4, 1.6, 2.0, 2.5, 3.0, 4.0, 5.5 };
    for (sigmas, 0..) |sigma_c, si| {
        const sigma = runtime(sigma_c);
        const radius: usize = @ceil(3.0 * sigma); // as the reference writes it (src/image.zig:973)
        const kernel_size = 2 * radius + 1;
        var kernel: [64]f32 = undefined;
        var sum: f32 = 0;
        for (0..kernel_size) |i| {
            const x = @as(f32, @floatFromInt(i)) - @as(f32, @floatFromInt(radius));
            kernel[i] = @exp(-(x * x) / (2.0 * sigma * sigma));
            sum += kernel[i];
        }
        for (kernel[0..kernel_size]) |*k| k.* /= sum;
        try w.print("{s}\"{d}\": [", .{ if (si == 0) "" else ", ", bits(sigma) });
        for (kernel[0..kernel_size], 0..) |k, i| try w.print("{s}{d}", .{ if (i == 0) "" else ",", bits(k) });
        try w.print("]", .{});
    }

    // src/image/channel_ops.zig:456-466: x_ratio in f32, src_x_f = (c + 0.5) * x_ratio - 0.5, fx = src_x_f - floor, weight k = kernel((k - 2) - fx)
    try w.print("}},\n  \"lanczos_plane_weights\": {{", .{});
    const geometries = [_][2]u32{ .{ 4096, 1500 }, .{ 640, 1000 } };
    for (geometries, 0..) |g, gi| {
        const ratio = runtime(@as(f32, @floatFromInt(g[0]))) / @as(f32, @floatFromInt(g[1]));
        try w.print("{s}\"{d}x{d}\": [", .{ if (gi == 0) "" else ", ", g[0], g[1] });
        for (0..g[1]) |c| {
            const src_x_f = (@as(f32, @floatFromInt(c)) + 0.5) * ratio - 0.5;
            const fx = src_x_f - @floor(src_x_f);
            for (0..6) |k| {
                const wk = lanczosPlaneKernel(@as(f32, @floatFromInt(@as(isize, @intCast(k)) - 2)) - fx);
                try w.print("{s}{d}", .{ if (c == 0 and k == 0) "" else ",", bits(wk) });
            }
        }
        try w.print("]", .{});
    }

    try w.print("}},\n  \"oklab_17\": [", .{});
    for (0..17) |ri| for (0..17) |gi| for (0..17) |bi| {
        const r: u8 = @intCast(@min(ri * 16, 255));
        const g: u8 = @intCast(@min(gi * 16, 255));
        const b: u8 = @intCast(@min(bi * 16, 255));
        const lab = xyzToOklab(rgbToXyz(r, g, b));
        try w.print("{s}{d},{d},{d}", .{ if (ri + gi + bi == 0) "" else ",", bits(lab[0]), bits(lab[1]), bits(lab[2]) });
    };
    try w.print("],\n  \"lab_17\": [", .{});
    for (0..17) |ri| for (0..17) |gi| for (0..17) |bi| {
        const r: u8 = @intCast(@min(ri * 16, 255));
        const g: u8 = @intCast(@min(gi * 16, 255));
        const b: u8 = @intCast(@min(bi * 16, 255));
        const lab = xyzToLab(rgbToXyz(r, g, b));
        try w.print("{s}{d},{d},{d}", .{ if (ri + gi + bi == 0) "" else ",", bits(lab[0]), bits(lab[1]), bits(lab[2]) });
    };
    try w.print("],\n", .{});

    // the argument ranges the path uses: Gaussian tap exponents; rotation angles and pi * x of the Lanczos kernels; LMS values;
    // (c + 0.055) / 1.055 of gammaToLinear; t of labForward; c of linearToGamma
    try sweep(w, "exp", fExp, 11, -90.0, 0.0, 4096, false);
    try sweep(w, "sin", fSin, 12, -12.0, 12.0, 4096, false);
    try sweep(w, "cos", fCos, 13, -12.0, 12.0, 4096, false);
    try sweep(w, "cbrt", fCbrt, 14, 0.0, 1.2, 4096, false);
    try sweep(w, "pow24", fPow24, 15, 0.0404, 1.0, 4096, false);
    try sweep(w, "pow_third", fPowThird, 16, 0.008856, 1.1, 4096, false);
    try sweep(w, "pow_inv24", fPowInv24, 17, 0.0031308, 1.0, 4096, true);
    try w.print("}}\n", .{});
    try w.flush();
}
