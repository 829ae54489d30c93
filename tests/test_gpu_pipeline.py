"""zg_batch_pipeline: the CLI's `pipeline` command (reference src/cli/pipeline.zig:153-179) over a batch of frames, against the
per-frame Image methods (which the other GPU tests hold to the oracle) and, for the BASELINE recipes, against the oracle itself."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import zignal_amd as zg  # noqa: E402
from zignal_amd import _lib as L  # noqa: E402

pytestmark = pytest.mark.gpu
I = zg.Interpolation


def frames_u8(oracle, seed, n, rows, cols, ch=4):
    shape = (n, rows, cols) if ch == 1 else (n, rows, cols, ch)
    return np.stack([oracle.synth_u8(seed + i, shape[1:]) for i in range(n)])


def per_frame(host, fn):
    return np.stack([fn(zg.Image(torch.from_numpy(f).cuda())).to_numpy() for f in host])


def check(host, steps, ref, what, space=None):
    got = zg.Pipeline(steps).run(torch.from_numpy(host).cuda(), space=space)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    want = per_frame(host, ref)
    assert got.shape == want.shape and got.dtype == want.dtype, (what, got.shape, want.shape)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"{what}: batch differs from the per-frame calls"
    return got


def test_blur_resize_both_orders_against_the_oracle(oracle):
    """the recipe [blur gaussian 0.6, resize 0.5 bilinear] of BASELINE configs[4] and its secondary order [resize, blur]."""
    host = frames_u8(oracle, 5, 5, 270, 480)
    bil = oracle.method(oracle.BILINEAR)
    got = check(host, [zg.Step.gaussian_blur(0.6), zg.Step.resize(135, 240)], lambda im: im.gaussian_blur(0.6).resize((135, 240), I.bilinear), "[blur, resize]")
    for f in range(host.shape[0]):
        assert np.array_equal(got[f], oracle.resize(oracle.gaussian_blur(host[f], 0.6), (135, 240), bil))
    got = check(host, [zg.Step.resize(135, 240), zg.Step.gaussian_blur(0.6)], lambda im: im.resize((135, 240), I.bilinear).gaussian_blur(0.6), "[resize, blur]")
    for f in range(host.shape[0]):
        assert np.array_equal(got[f], oracle.gaussian_blur(oracle.resize(host[f], (135, 240), bil), 0.6))


def test_resize_convert_config3_recipe(oracle):
    """BASELINE configs[2] as a recipe: resize(.bilinear) 4:1 then convert(Oklab f32), fused over the batch."""
    host = frames_u8(oracle, 3, 3, 512, 768)
    got = check(host, [zg.Step.resize(128, 192), zg.Step.convert(zg.CS_OKLAB)], lambda im: im.resize((128, 192), I.bilinear).convert(zg.CS_OKLAB, np.float32), "[resize, convert]")
    want0 = oracle.convert(oracle.resize(host[0], (128, 192), oracle.method(oracle.BILINEAR)), oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3)
    assert np.array_equal(got[0].view(np.uint32), want0.view(np.uint32))
    check(host, [zg.Step.resize(128, 192), zg.Step.convert(zg.CS_XYZ)], lambda im: im.resize((128, 192), I.bilinear).convert(zg.CS_XYZ, np.float32), "[resize, convert xyz]")
    check(host, [zg.Step.resize(100, 333, I.bicubic), zg.Step.convert(zg.CS_LAB)], lambda im: im.resize((100, 333), I.bicubic).convert(zg.CS_LAB, np.float32), "[resize bicubic, convert lab]")


@pytest.mark.parametrize("n", (1, 4, 9))
def test_long_recipes_mixed_steps(oracle, n):
    """Four and five steps, batched and per-frame kernels mixed, shapes and types changing along the way (ping-pong scratch)."""
    host = frames_u8(oracle, 40, n, 96, 160)
    tr = zg.AffineTransform(np.array([[0.9, 0.1], [-0.1, 0.9]], np.float32), np.array([3.0, -2.0], np.float32))
    steps = [zg.Step.gaussian_blur(1.0), zg.Step.resize(120, 200, I.bilinear), zg.Step.box_blur(2), zg.Step.warp(tr, 64, 80, I.bicubic), zg.Step.convert(zg.CS_OKLAB)]
    ref = lambda im: im.gaussian_blur(1.0).resize((120, 200), I.bilinear).box_blur(2).warp(tr, (64, 80), I.bicubic).convert(zg.CS_OKLAB, np.float32)  # noqa: E731
    check(host, steps, ref, f"five steps, {n} frames")
    check(host, steps[:4], lambda im: im.gaussian_blur(1.0).resize((120, 200), I.bilinear).box_blur(2).warp(tr, (64, 80), I.bicubic), f"four steps, {n} frames")
    check(host, [zg.Step.convert(zg.CS_GRAY, np.uint8), zg.Step.gaussian_blur(2.5), zg.Step.resize(48, 80, I.nearest)],
          lambda im: im.convert(zg.CS_GRAY, np.uint8).gaussian_blur(2.5).resize((48, 80), I.nearest), f"grey long blur, {n} frames")


def test_other_pixel_types_and_single_steps(oracle):
    f32 = np.stack([oracle.synth_f32(70 + i, (66, 130, 4)) for i in range(3)])
    check(f32, [zg.Step.gaussian_blur(0.6), zg.Step.resize(33, 65)], lambda im: im.gaussian_blur(0.6).resize((33, 65), I.bilinear), "Rgba(f32) [blur, resize]")
    rgb = frames_u8(oracle, 80, 4, 100, 256, 3)
    check(rgb, [zg.Step.gaussian_blur(0.6)], lambda im: im.gaussian_blur(0.6), "Rgb(u8) blur")
    check(rgb, [zg.Step.resize(50, 128), zg.Step.convert(zg.CS_HSV)], lambda im: im.resize((50, 128), I.bilinear).convert(zg.CS_HSV, np.float32), "Rgb(u8) [resize, convert hsv]")
    grey = frames_u8(oracle, 90, 6, 128, 256, 1)
    check(grey, [zg.Step.gaussian_blur(0.6), zg.Step.gaussian_blur(0.0)], lambda im: im.gaussian_blur(0.6), "grey blur + sigma 0 copy")
    rgba = frames_u8(oracle, 95, 2, 64, 64)
    check(rgba, [], lambda im: im, "no steps: copy")
    check(rgba, [zg.Step.resize(64, 64)], lambda im: im, "equal-size resize: copy")
    check(rgba, [zg.Step.resize(128, 256)], lambda im: im.resize((128, 256), I.bilinear), "upscale (four pixels per lane)")


def test_validation_happens_before_any_work():
    lib = zg.lib()
    p = zg.Pipeline([zg.Step.gaussian_blur(0.6), zg.Step.gaussian_blur(-1.0)])
    with pytest.raises(zg.InvalidArgument):
        p.out_layout(10, 10, L.PIXEL_RGBA_U8, L.CS_RGBA)
    bad = L.ZgStep()
    bad.kind = 17
    arr = (L.ZgStep * 1)(bad)
    assert lib.zg_batch_pipeline_shape(4, 4, L.PIXEL_U8, L.CS_GRAY, arr, 1, None, None, None, None) == L.ERR_INVALID_ARGUMENT
    t = torch.zeros((2, 8, 8, 4), dtype=torch.uint8, device="cuda")
    with pytest.raises(zg.ZignalError):
        zg.Pipeline([zg.Step.convert(zg.CS_OKLAB, np.uint8)]).run(t)  # float-only colour type into u8 pixels
    with pytest.raises(ValueError):
        zg.Pipeline([]).run(torch.zeros((2, 8, 8, 4), dtype=torch.uint8))  # host tensor


def test_large_batch_goes_through_in_groups(oracle):
    """Intermediates above the scratch budget: 40 frames of 1080p through three steps run as several groups, same bits."""
    one = oracle.synth_u8(123, (1080, 1920, 4))
    host = np.stack([np.roll(one, 7 * i, axis=1) for i in range(40)])
    got = zg.Pipeline([zg.Step.gaussian_blur(0.6), zg.Step.resize(540, 960), zg.Step.gaussian_blur(0.6)]).run(torch.from_numpy(host).cuda())
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for f in (0, 17, 39):
        want = zg.Image(torch.from_numpy(host[f]).cuda()).gaussian_blur(0.6).resize((540, 960), I.bilinear).gaussian_blur(0.6).to_numpy()
        assert np.array_equal(got[f], want), f


def _edges_bridge(oracle, frame, detector):
    """edges.apply (src/cli/edges.zig:126-135) with the oracle: convert(u8) -> detector -> convert(Rgba(u8)), three separate calls."""
    grey = oracle.convert(frame, oracle.CS_RGBA, oracle.CS_GRAY, np.uint8, 1)
    return oracle.convert(detector(grey), oracle.CS_GRAY, oracle.CS_RGBA, np.uint8, 4)


def test_the_reference_example_recipe_resize_lanczos_blur_sobel(oracle):
    """The recipe in the CLI's own help text (src/cli/pipeline.zig:58-67): resize with .lanczos, blur gaussian sigma 2, edges sobel — one
    zg_batch_pipeline call over the batch against the oracle doing the calls one by one."""
    host = frames_u8(oracle, 71, 6, 120, 200)
    steps = [zg.Step.resize(90, 150, I.lanczos), zg.Step.gaussian_blur(2.0), zg.Step.edges_sobel()]
    got = zg.Pipeline(steps).run(torch.from_numpy(host).cuda())
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == (6, 90, 150, 4) and got.dtype == np.uint8
    lan = oracle.method(oracle.LANCZOS)
    for f in range(host.shape[0]):
        want = _edges_bridge(oracle, oracle.gaussian_blur(oracle.resize(host[f], (90, 150), lan), 2.0), oracle.sobel)
        assert np.array_equal(got[f], want), f"frame {f}"


def test_the_reference_example_recipe_on_64_frames_of_1080p(oracle):
    """VERDICT r03 item 4's bar: the same recipe over 64 x 1080p frames (the CLI's --width 800 keeps the aspect: 450 x 800); every
    frame against the per-frame device calls, a sample of frames against the oracle."""
    n = 64
    rng = np.random.default_rng(9)
    base = frames_u8(oracle, 90, 4, 1080, 1920)
    host = np.stack([np.roll(base[i % 4], (7 * i, 13 * i), axis=(0, 1)) ^ np.uint8(i) for i in range(n)])
    steps = [zg.Step.resize(450, 800, I.lanczos), zg.Step.gaussian_blur(2.0), zg.Step.edges_sobel()]
    dev = torch.from_numpy(host).cuda()
    got = zg.Pipeline(steps).run(dev)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    grey = lambda im: im.resize((450, 800), I.lanczos).gaussian_blur(2.0).convert(zg.CS_GRAY, np.uint8).sobel().convert(zg.CS_RGBA, np.uint8, src_space=zg.CS_GRAY)
    for f in range(n):
        want = grey(zg.Image(dev[f])).to_numpy()
        assert np.array_equal(got[f], want), f"frame {f} differs from the per-frame calls"
    lan = oracle.method(oracle.LANCZOS)
    for f in (0, 17, 63):
        want = _edges_bridge(oracle, oracle.gaussian_blur(oracle.resize(host[f], (450, 800), lan), 2.0), oracle.sobel)
        assert np.array_equal(got[f], want), f"frame {f} differs from the oracle"


@pytest.mark.parametrize("ch", (1, 3, 4))
def test_long_tap_blur_and_plane_resizers_run_with_the_frame_in_the_grid(oracle, ch):
    """The two-pass u16 Gaussian (conv_sep_bytes2.hip) and the Rgb(u8) / Rgba(u8) plane resizers (resize_planes.hip) take the whole batch
    per launch (frame = blockIdx.y): the same bits as the per-frame calls for every method and for sigmas on both sides of the u16 temp's
    clamp, and frame 0 against the oracle."""
    cols = {1: 512, 3: 336, 4: 192}[ch]  # row bytes a multiple of 16 and >= 256: the two-pass kernels' precondition
    host = frames_u8(oracle, 300 + ch, 5, 77, cols, ch)
    for sigma in (1.6, 2.0, 4.5):
        got = check(host, [zg.Step.gaussian_blur(sigma)], lambda im: im.gaussian_blur(sigma), f"blur {sigma} ch {ch}")
        assert np.array_equal(got[0], oracle.gaussian_blur(host[0], sigma))
    if ch == 1:
        return
    for m, om in ((I.nearest, oracle.NEAREST), (I.bilinear, oracle.BILINEAR), (I.bicubic, oracle.BICUBIC), (I.catmull_rom, oracle.CATMULL_ROM),
                  (I.mitchell_default, oracle.MITCHELL), (I.lanczos, oracle.LANCZOS)):
        for shape in ((40, 100), (131, 333)):
            got = check(host, [zg.Step.resize(*shape, m)], lambda im: im.resize(shape, m), f"resize {m} {shape} ch {ch}")
            assert np.array_equal(got[0], oracle.resize(host[0], shape, oracle.method(om)))


def test_every_blur_type_and_edge_detector_of_the_cli_as_a_step(oracle):
    """blur: box, gaussian, median, motion_linear, motion_zoom, motion_spin (src/cli/blur.zig:98-170); edges: sobel, canny, shen_castan
    through the grey bridge (src/cli/edges.zig:85-135) — each as one step over a batch, against the oracle per frame."""
    host = frames_u8(oracle, 23, 3, 72, 104)
    cs = oracle.cos_sin(0.6)  # one pair for both sides: the angle's cos / sin are the caller's (a Zig host brings Zig's)
    cases = [
        ("median r=1", zg.Step.median_blur(1), lambda a: oracle.order_statistic_blur(a, 1, 0, 0.5)),
        ("median r=2", zg.Step.median_blur(2), lambda a: oracle.order_statistic_blur(a, 2, 0, 0.5)),
        ("motion linear", zg.Step.motion_blur_linear(0.6, 7, cos_sin=cs), lambda a: oracle.motion_blur_linear(a, 0.6, 7, cos_sin=cs)),
        ("motion linear axis", zg.Step.motion_blur_linear(0.0, 9, cos_sin=(1.0, 0.0)), lambda a: oracle.motion_blur_linear(a, 0.0, 9, cos_sin=(1.0, 0.0))),
        ("motion zoom", zg.Step.motion_blur_radial(0.4, 0.6, 0.5), lambda a: oracle.motion_blur_radial(a, 0.4, 0.6, 0.5, False)),
        ("motion spin", zg.Step.motion_blur_radial(0.5, 0.5, 0.3, spin=True), lambda a: oracle.motion_blur_radial(a, 0.5, 0.5, 0.3, True)),
        ("sobel", zg.Step.edges_sobel(), lambda a: _edges_bridge(oracle, a, oracle.sobel)),
        ("canny", zg.Step.edges_canny(1.0, 50.0, 100.0), lambda a: _edges_bridge(oracle, a, lambda g: oracle.canny(g, 1.0, 50.0, 100.0))),
        ("shen-castan", zg.Step.edges_shen_castan(), lambda a: _edges_bridge(oracle, a, oracle.shen_castan)),
        ("shen-castan nms", zg.Step.edges_shen_castan(0.8, 9, 0.95, 0.4, True), lambda a: _edges_bridge(oracle, a, lambda g: oracle.shen_castan(g, 0.8, 9, 0.95, 0.4, True, True))),
    ]
    dev = torch.from_numpy(host).cuda()
    for what, step, ref in cases:
        got = zg.Pipeline([step]).run(dev)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        assert got.shape == host.shape and got.dtype == np.uint8, what
        for f in range(host.shape[0]):
            assert np.array_equal(got[f], ref(host[f])), f"{what}: frame {f}"
    # grey and Rgb frames keep their type through the bridge; an edges step in the middle of a recipe
    grey = frames_u8(oracle, 31, 2, 64, 96, ch=1)
    got = zg.Pipeline([zg.Step.edges_sobel()]).run(torch.from_numpy(grey).cuda()).cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], oracle.sobel(grey[f]))
    got = zg.Pipeline([zg.Step.gaussian_blur(1.0), zg.Step.edges_canny(1.4, 40.0, 120.0), zg.Step.resize(36, 52)]).run(dev).cpu().numpy()
    bil = oracle.method(oracle.BILINEAR)
    for f in range(host.shape[0]):
        want = oracle.resize(_edges_bridge(oracle, oracle.gaussian_blur(host[f], 1.0), lambda g: oracle.canny(g, 1.4, 40.0, 120.0)), (36, 52), bil)
        assert np.array_equal(got[f], want), f"[blur, canny, resize]: frame {f}"


@pytest.mark.parametrize("ch", (1, 3, 4))
def test_box_blur_step_takes_the_batch_in_the_grid(oracle, ch):
    """boxBlur as a pipeline step (src/cli/blur.zig:109-112): the three kernels (strip carries, SAT chain, window means) take the whole batch per
    launch, frame index in the grid, in groups whose integral images fit a scratch block — against Image.boxBlur of the oracle frame by frame,
    bright frames included (SAT values past 2^24: the column recurrence's rounding order is what is checked)."""
    rng = np.random.default_rng(30 + ch)
    shape = (7, 200, 328) + ((ch,) if ch > 1 else ())
    host = rng.integers(0, 256, shape, dtype=np.uint8)
    host[3] = 255 - (host[3] >> 5)  # a bright frame
    dev = torch.from_numpy(host).cuda()
    for radius in (1, 2, 9):
        got = zg.Pipeline([zg.Step.box_blur(radius)]).run(dev)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        for f in range(shape[0]):
            assert np.array_equal(got[f], oracle.box_blur(host[f], radius)), f"box r={radius} ch={ch}: frame {f}"
    f32 = rng.random((3, 64, 96), dtype=np.float32)  # f32 frames: the per-image SAT kernels, frame by frame
    got = zg.Pipeline([zg.Step.box_blur(2)]).run(torch.from_numpy(f32).cuda()).cpu().numpy()
    for f in range(3):
        assert np.array_equal(got[f].view(np.uint32), oracle.box_blur(f32[f], 2).view(np.uint32)), f"box f32: frame {f}"


def test_edges_step_bridges_float_frames_and_other_colour_spaces(oracle):
    """edges.apply is frame.convert(u8) -> detector -> .convert(frame type) (src/cli/edges.zig:126-135). Float frames must go through that
    conversion (Image(f32).sobel on raw [0, 1] floats is a different picture), and so must frames an earlier CONVERT step left in Oklab."""
    rng = np.random.default_rng(9)
    f32 = rng.random((3, 48, 80), dtype=np.float32)
    for step, det in ((zg.Step.edges_sobel(), oracle.sobel), (zg.Step.edges_canny(1.0, 50.0, 100.0), lambda g: oracle.canny(g, 1.0, 50.0, 100.0))):
        got = zg.Pipeline([step]).run(torch.from_numpy(f32).cuda())
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        assert got.dtype == np.float32 and got.shape == f32.shape
        for f in range(3):
            g8 = oracle.convert(f32[f], oracle.CS_GRAY, oracle.CS_GRAY, np.uint8, 1)
            want = oracle.convert(det(g8), oracle.CS_GRAY, oracle.CS_GRAY, np.float32, 1)
            assert np.array_equal(got[f].view(np.uint32), want.view(np.uint32)), f"f32 grey frame {f}"
        assert got.max() > 0.2, "edges of a [0, 1] float image must not vanish"
    rgbf = rng.random((2, 40, 64, 3), dtype=np.float32)
    got = zg.Pipeline([zg.Step.edges_sobel()]).run(torch.from_numpy(rgbf).cuda()).cpu().numpy()
    for f in range(2):
        g8 = oracle.convert(rgbf[f], oracle.CS_RGB, oracle.CS_GRAY, np.uint8, 1)
        want = oracle.convert(oracle.sobel(g8), oracle.CS_GRAY, oracle.CS_RGB, np.float32, 3)
        assert np.array_equal(got[f].view(np.uint32), want.view(np.uint32)), f"Rgb(f32) frame {f}"
    host = frames_u8(oracle, 41, 2, 40, 64)
    got = zg.Pipeline([zg.Step.convert(zg.CS_OKLAB), zg.Step.edges_sobel()]).run(torch.from_numpy(host).cuda()).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == (2, 40, 64, 3)
    for f in range(2):
        lab = oracle.convert(host[f], oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3)
        g8 = oracle.convert(lab, oracle.CS_OKLAB, oracle.CS_GRAY, np.uint8, 1)
        want = oracle.convert(oracle.sobel(g8), oracle.CS_GRAY, oracle.CS_OKLAB, np.float32, 3)
        assert np.array_equal(got[f].view(np.uint32), want.view(np.uint32)), f"Oklab frame {f}"


def test_run_multi_equals_run_on_the_devices_this_box_has(oracle):
    """zg_multi_batch_pipeline behind Pipeline.run_multi: any recipe over a zg_multi context. One GPU here, so world 1 (the root works in place);
    the C++ program (tests/cpp/test_device_image.cpp) adds the RCCL loop-back and, where two GPUs are visible, the world > 1 branches."""
    host = frames_u8(oracle, 61, 7, 60, 88)
    dev = torch.from_numpy(host).cuda()
    recipes = ([zg.Step.gaussian_blur(0.6), zg.Step.resize(30, 44)], [zg.Step.resize(45, 66, I.bicubic), zg.Step.gaussian_blur(1.5), zg.Step.convert(zg.CS_OKLAB)],
               [zg.Step.box_blur(2), zg.Step.edges_sobel()])
    with zg.Multi([0]) as ctx:
        assert ctx.device_count() == 1
        for steps in recipes:
            p = zg.Pipeline(steps)
            want = p.run(dev)
            got, times = p.run_multi(ctx, dev)
            assert torch.equal(got, want)
            assert len(times) == 3 and times[2] > 0
        with pytest.raises(zg.InvalidArgument):
            zg.Pipeline([zg.Step.gaussian_blur(-1.0)]).run_multi(ctx, dev)
        got, _ = zg.Pipeline(recipes[0]).run_multi(ctx, dev)  # an argument error leaves the context usable
        assert torch.equal(got, zg.Pipeline(recipes[0]).run(dev))


def test_new_steps_are_validated_before_anything_runs():
    dev = torch.zeros((2, 32, 32, 4), dtype=torch.uint8, device="cuda")
    bad = zg.Step.edges_sobel()
    bad.c.edges = 7
    with pytest.raises(zg.InvalidArgument):
        zg.Pipeline([zg.Step.gaussian_blur(1.0), bad]).run(dev)
    bad = zg.Step.motion_blur_radial()
    bad.c.motion = -1
    with pytest.raises(zg.InvalidArgument):
        zg.Pipeline([bad]).run(dev)
    with pytest.raises(zg.InvalidArgument):
        zg.Pipeline([zg.Step.median_blur(300)]).run(dev)
