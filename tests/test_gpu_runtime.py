"""Runtime services behind the C ABI: scratch ownership under graph capture, the idle-scratch cache and its trim, and the role-split
SAT chain against the plain two-kernel form on awkward shapes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import zignal_amd as zg  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_a_graph_owns_the_scratch_its_capture_took():
    """zg_graph_end_capture hands the capture's scratch blocks to the graph and zg_graph_destroy frees them: re-capturing in a loop
    does not grow the device's footprint, and zg_release_graph_scratch (for foreign captures) leaves living graphs alone."""
    lib = zg.lib()
    src = zg.Image(torch.randint(0, 256, (2048, 2048, 4), dtype=torch.uint8, device="cuda"))
    dst = zg.Image(torch.empty_like(src.data))
    stream = torch.cuda.Stream()
    want = src.gaussian_blur(2.5).to_numpy()  # 17 taps: the two-pass path with a 32 MiB u16 temp plane
    lib.zg_trim_scratch()
    base = free_bytes()
    graphs = []
    for i in range(6):
        with torch.cuda.stream(stream):
            assert lib.zg_graph_begin_capture(C.c_void_p(stream.cuda_stream)) == 0
            src.gaussian_blur(2.5, out=dst)
            g = C.c_void_p()
            assert lib.zg_graph_end_capture(C.c_void_p(stream.cuda_stream), C.byref(g)) == 0, lib.zg_last_error()
        graphs.append(g)
        if i == 2:  # a live graph survives somebody else's clean-up call
            assert lib.zg_release_graph_scratch() == 0
        with torch.cuda.stream(stream):
            dst.data.zero_()  # on the stream the graph replays on
        assert lib.zg_graph_launch(g, C.c_void_p(stream.cuda_stream)) == 0
        stream.synchronize()
        assert np.array_equal(dst.to_numpy(), want), f"replay of capture {i}"
    held = base - free_bytes()
    assert held >= 6 * 30 * 2**20, held  # six graphs, six temp planes
    for g in graphs[:3]:
        assert lib.zg_graph_destroy(g) == 0
    assert base - free_bytes() <= held // 2 + 8 * 2**20
    with torch.cuda.stream(stream):
        dst.data.zero_()
    assert lib.zg_graph_launch(graphs[4], C.c_void_p(stream.cuda_stream)) == 0  # the others still work
    stream.synchronize()
    assert np.array_equal(dst.to_numpy(), want)
    for g in graphs[3:]:
        assert lib.zg_graph_destroy(g) == 0
    assert base - free_bytes() <= 8 * 2**20


def test_trim_returns_the_idle_scratch_cache():
    lib = zg.lib()
    host = np.random.default_rng(0).integers(0, 256, (4096, 4096, 4), dtype=np.uint8)
    zg.Image(host).box_blur(2)  # once, so that what the HIP runtime allocates on first use (code objects, staging) is in the baseline
    lib.zg_trim_scratch()
    base = free_bytes()
    zg.Image(host).box_blur(2)  # host-pointer layer: the frame's device twins and the SAT come from the cache
    assert base - free_bytes() >= 64 * 2**20
    assert lib.zg_trim_scratch() == 0
    assert base - free_bytes() <= 8 * 2**20
    # a capped cache keeps nothing above its budget (the cap is read once per process: a child process)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import zignal_amd as zg\n"
            "h = np.zeros((4096, 4096, 4), np.uint8); zg.Image(h).box_blur(2); zg.lib().zg_trim_scratch(); torch.cuda.synchronize()\n"
            "b = torch.cuda.mem_get_info()[0]; zg.Image(h).box_blur(2); torch.cuda.synchronize(); print(b - torch.cuda.mem_get_info()[0])" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, ZIGNAL_HIP_SCRATCH_CACHE_MB="16"))
    assert out.returncode == 0, out.stderr[-1500:]
    assert int(out.stdout.strip().splitlines()[-1]) <= 24 * 2**20, out.stdout  # 16 MiB budget + allocation granularity


@pytest.mark.parametrize("shape", ((63, 97), (64, 64), (1, 300), (300, 1), (65, 4097), (130, 1023, 4), (67, 129, 3), (257, 63, 4)))
def test_role_split_sat_chain_equals_the_unfused_kernels(shape, oracle):
    """k_sat_chain retires some of its waves before the others reach their barriers (fine on CDNA, outside HIP's barrier contract):
    hold it to the plain row-scan + column-scan pair (ZIGNAL_HIP_SAT_UNFUSED=1, a child process) and to the oracle on shapes with
    rows / columns off the 64 grid, single rows and columns, and strided views."""
    host = oracle.synth_u8(500 + shape[0], shape)
    view = oracle.synth_u8(900 + shape[0], (shape[0] + 5, shape[1] + 9) + shape[2:])
    here = {}
    for name, arr in (("whole", host), ("view", view)):
        im = zg.Image(torch.from_numpy(arr).cuda())
        if name == "view":
            im = im.view((3, 2, 3 + shape[1], 2 + shape[0]))
        here[name] = [im.box_blur(r).to_numpy() for r in (1, 2, 7)] + [im.sharpen(2).to_numpy()]
    src_whole, src_view = host, view[2:2 + shape[0], 3:3 + shape[1]]
    for r, got in zip((1, 2, 7), here["whole"]):
        assert np.array_equal(got, oracle.box_blur(src_whole, r)), (shape, r)
    for r, got in zip((1, 2, 7), here["view"]):
        assert np.array_equal(got, oracle.box_blur(np.ascontiguousarray(src_view), r)), (shape, r, "view")
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); import zignal_amd as zg\n"
            "a = np.load(sys.argv[1]); im = zg.Image(torch.from_numpy(a).cuda())\n"
            "np.savez(sys.argv[2], *([im.box_blur(r).to_numpy() for r in (1, 2, 7)] + [im.sharpen(2).to_numpy()]))" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "in.npy"), host)
        out = subprocess.run([sys.executable, "-c", code, os.path.join(d, "in.npy"), os.path.join(d, "out.npz")], capture_output=True, text=True,
                             timeout=300, env=dict(os.environ, ZIGNAL_HIP_SAT_UNFUSED="1"))
        assert out.returncode == 0, out.stderr[-1500:]
        other = np.load(os.path.join(d, "out.npz"))
        for i, got in enumerate(here["whole"]):
            assert np.array_equal(got, other[f"arr_{i}"]), (shape, i)


ALTERNATIVE_FORMS = {
    # hook: what it switches back to. Round 4 dropped the hooks of the forms that had lost by more than a tenth and whose numbers are on file
    # (profiles/r03_experiments.txt, r04_experiments.txt): NO_STREAM, STREAM_GREY, ROWS_INT, COLS_INT, NO_LAB4, NO_U8_PLANE_RESIZE, NO_WARP_STAGE, CONV2D_INT.
    # The forms themselves stay where ordinary inputs still reach them (shapes a fast kernel's preconditions exclude) and are tested there.
    # Round 5 dropped NO_CONV2D_STREAM, NO_SOBEL_STREAM (27 / 20 us against 75 / 55: r04_experiments.txt), the strip-height knobs and MFMA (the matrix-pipe
    # Gaussian left the library: tools/exp/conv_sep_mfma.hip), and added this round's eight.
    "ZIGNAL_HIP_STREAM_NO_FOLD": "k_sep_stream's plain row pass and end-tap multiplies instead of the folded unit-end form (gaussianBlur(0.6)'s taps)",
    "ZIGNAL_HIP_NO_TILE_F32": "the LDS-tiled k_sep_f32x4 instead of the tile-per-wave k_sep_tile_f32 for Image(f32) planes",
    "ZIGNAL_HIP_RESIZE_FORM=0": "round 4's four-row workgroups in XCD-major order for every bilinear Rgba(u8) resize (reductions use one-wave workgroups in address order)",
    "ZIGNAL_HIP_NO_PYRAMID_FUSE": "gaussianBlur into a blurred plane then resize for every level of an Image(u8) pyramid instead of the column pass fused with the bilinear taps",
    "ZIGNAL_HIP_NO_PYRAMID_BATCH": "every long-tap level of an Image(u8) pyramid with its own row- and column-pass launches instead of the multi-job kernels (k_rows_u8f_multi, k_cols_u8f_multi, k_cols_bilinear_u8_multi)",
    "ZIGNAL_HIP_RESIZE_U8_ROWS=0": "k_resize_bilinear_u8 (one output row per wave, taps per pixel) for every Image(u8) bilinear resize instead of k_resize_bilinear_u8_rows<2> below a ratio of 2",
    "ZIGNAL_HIP_RESIZE_U8_ROWS=4": "k_resize_bilinear_u8_rows<4>: four output rows per wave",
    "ZIGNAL_HIP_B2_GENERIC_COLS": "k_cols_u8f (tap count at run time) for every band instead of k_cols_u8f_static<NK4> (unrolled, taps in SGPRs) in the long-tap u8 column pass",
    "ZIGNAL_HIP_KEEP_ZERO_TAPS": "integer kernels with their zero outer taps kept (gaussianBlur's 3-sigma radius rounds them to zero from sigma 2.3 up)",
    "ZIGNAL_HIP_ISEF_TRANSPOSE": "two transposes around k_isef_cols instead of the recursions along the rows",
    "ZIGNAL_HIP_ISEF_SERIAL": "the role-split k_isef (one chain per row / column from end to end) instead of the segmented k_isef_spec",
    "ZIGNAL_HIP_SC_THREE_SATS": "shenCastan's window count from the mask's integral image (three SATs, twelve corner loads) instead of k_sc_count",
    "ZIGNAL_HIP_ISEF_W=4": "k_isef_spec with a four-step warm-up: segments start wrong all over the plane and the repair launch redoes them",
}


@pytest.mark.parametrize("hook", sorted(ALTERNATIVE_FORMS))
def test_the_alternative_kernel_forms_behind_the_tuning_hooks_give_the_same_bits(hook):
    """Every ZIGNAL_HIP_* hook that swaps one kernel form for another (they exist for A/B timing) must leave the results
    untouched: the same calls, against the oracle, with the hook set (a child process: the hooks are read once)."""
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import zignal_amd as zg
from oracle import pyoracle as o
o.lib()
def dev(a): return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())
def same(got, want, what):
    torch.cuda.synchronize()
    g = got.to_numpy()
    assert g.shape == want.shape and g.dtype == want.dtype, what
    assert np.array_equal(g.view(np.uint8), want.view(np.uint8)), what
rng = np.random.default_rng(5)
rgba = rng.integers(0, 256, (70, 1100, 4), dtype=np.uint8)
grey = rng.integers(0, 256, (300, 1296), dtype=np.uint8)
same(dev(rgba).convert(zg.CS_OKLAB, np.float32), o.convert(rgba, o.CS_RGBA, o.CS_OKLAB, np.float32, 3), "oklab")
same(dev(rgba[..., :3]).convert(zg.CS_XYZ, np.float32), o.convert(np.ascontiguousarray(rgba[..., :3]), o.CS_RGB, o.CS_XYZ, np.float32, 3), "xyz")
lab = o.convert(rgba, o.CS_RGBA, o.CS_LAB, np.float32, 3)
same(dev(rgba).convert(zg.CS_LAB, np.float32), lab, "lab")
same(dev(lab).convert(zg.CS_RGBA, np.uint8, src_space=zg.CS_LAB), o.convert(lab, o.CS_LAB, o.CS_RGBA, np.uint8, 4), "lab back")
for sigma in (0.6, 1.0, 2.25, 5.5):
    same(dev(grey).gaussian_blur(sigma), o.gaussian_blur(grey, sigma), "grey blur %%g" %% sigma)
    same(dev(rgba).gaussian_blur(sigma), o.gaussian_blur(rgba, sigma), "rgba blur %%g" %% sigma)
for size in ((250, 1080), (97, 411), (640, 2600)):
    same(dev(grey).resize(size, zg.Interpolation.bilinear), o.resize(grey, size, o.method(o.BILINEAR)), "grey resize")
f32 = rng.random((90, 700, 4), dtype=np.float32)
plane = rng.random((90, 700), dtype=np.float32) - np.float32(0.5)
for sigma in (0.3, 0.6, 1.0):
    same(dev(plane).gaussian_blur(sigma), o.gaussian_blur(plane, sigma), "f32 plane blur %%g" %% sigma)
for size in ((20, 300), (35, 550), (140, 2200)):
    same(dev(rgba).resize(size, zg.Interpolation.bilinear), o.resize(rgba, size, o.method(o.BILINEAR)), "rgba resize")
small = o.resize(rgba, (20, 300), o.method(o.BILINEAR))
same(dev(rgba).resize_convert((20, 300), zg.CS_OKLAB), o.convert(small, o.CS_RGBA, o.CS_OKLAB, np.float32, 3), "resize + oklab")
for size, kind, okind in (((120, 930), zg.Interpolation.bicubic, o.BICUBIC), ((45, 350), zg.Interpolation.catmull_rom, o.CATMULL_ROM), ((95, 705), zg.Interpolation.mitchell(1 / 3, 1 / 3), o.MITCHELL)):
    same(dev(f32).resize(size, kind), o.resize(f32, size, o.method(okind, kind.b, kind.c)), "f32 resize")
for n in (3, 5, 7):
    k = rng.random((n, n), dtype=np.float32) / (n * n)
    k[0, 0] *= -1
    same(dev(rgba).convolve(k), o.convolve(rgba, k, o.MIRROR), "convolve rgba")
    same(dev(grey).convolve(k), o.convolve(grey, k, o.MIRROR), "convolve grey")
same(dev(rgba).sobel(), o.sobel(rgba), "sobel rgba")
same(dev(grey).sobel(), o.sobel(grey), "sobel grey")
same(dev(grey).shen_castan(), o.shen_castan(grey), "shen-castan grey")
plane = rng.integers(0, 256, (333, 1296)).astype(np.float32)
for smooth in (0.95, 0.9, 0.7, 0.4):
    same(dev(plane).isef_smooth(smooth), o.isef_plane(plane, smooth), "isef %%g" %% smooth)
same(dev(rgba).shen_castan(smooth=0.6, use_nms=True), o.shen_castan(rgba, smooth=0.6, use_nms=True), "shen-castan rgba")
for got, want in zip(zg.ImagePyramid.build(dev(grey), 6, 1.2, 1.6).levels, o.pyramid(grey, 6, 1.2, 1.6)):
    same(got, want, "pyramid level")
for w in (3, 5, 11, 15):
    same(dev(grey).shen_castan(window_size=w, high_ratio=0.9), o.shen_castan(grey, window_size=w, high_ratio=0.9), "shen-castan window %%d" %% w)
print("ok")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **{hook.partition("=")[0]: hook.partition("=")[2] or "1"}))
    assert out.returncode == 0 and "ok" in out.stdout, (ALTERNATIVE_FORMS[hook], out.stdout[-400:], out.stderr[-1200:])
