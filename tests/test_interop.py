"""numpy / torch interop of the Python mirror against the reference binding's contract
(bindings/python/src/image/numpy_interop.zig:114-210): zero-copy, row strides allowed, pixels of a row contiguous,
TypeError for None or a wrong dtype, ValueError for a wrong shape or incompatible strides. CPU only."""
import numpy as np
import pytest

import zignal_amd as zg


def test_from_numpy_is_zero_copy_and_keeps_row_strides():
    a = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    im = zg.Image.from_numpy(a)
    assert (im.rows, im.cols, im.stride) == (5, 7, 7) and np.shares_memory(im.data, a)
    a[2, 3, 1] = 200
    assert im.to_numpy()[2, 3, 1] == 200  # same memory, both ways
    view = a[1:4, 2:6]                    # a view: the row stride stays the parent's (numpy_interop.zig:166-170)
    iv = zg.Image.from_numpy(view)
    assert (iv.rows, iv.cols, iv.stride) == (3, 4, 7) and np.shares_memory(iv.data, a) and not iv.is_contiguous()
    for shape, pixel in (((4, 6), 0), ((4, 6, 1), 0), ((4, 6, 3), 2), ((4, 6, 4), 3)):
        assert zg.Image.from_numpy(np.zeros(shape, np.uint8)).pixel == pixel
    assert zg.Image.from_numpy(np.zeros((4, 6, 4), np.float32)).pixel == 5
    assert zg.Image.from_numpy(np.zeros((4, 6, 1), np.uint8)).to_numpy().shape == (4, 6, 1)  # Gray -> (rows, cols, 1) survives


def test_from_numpy_errors_follow_the_reference():
    a = np.zeros((5, 7, 3), np.uint8)
    with pytest.raises(TypeError):
        zg.Image.from_numpy(None)
    with pytest.raises(TypeError):
        zg.Image.from_numpy(a.astype(np.int16))
    for bad in (a.transpose(1, 0, 2), a[:, ::2], a[..., ::2], np.zeros((5, 7, 2), np.uint8), np.zeros(5, np.uint8), np.zeros((2, 2, 2, 3), np.uint8)):
        with pytest.raises(ValueError):
            zg.Image.from_numpy(bad)
    zg.Image.from_numpy(np.ascontiguousarray(a.transpose(1, 0, 2)))  # the reference's advice for such layouts


def test_torch_tensors_are_device_memory_only():
    torch = pytest.importorskip("torch")
    with pytest.raises(ValueError):
        zg.Image(torch.zeros((4, 4, 3), dtype=torch.uint8))  # a CPU tensor: pass .numpy() (zero-copy) for host pixels
    t = torch.zeros((4, 4, 3), dtype=torch.uint8)
    im = zg.Image.from_numpy(t.numpy())
    t[1, 2, 0] = 9
    assert im.data[1, 2, 0] == 9


def test_host_images_need_the_library_not_a_fallback():
    """Host pixels go through zg_<op>_host: without a GPU the call must fail loudly, never compute on the CPU."""
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the loud failure cannot be observed")
    with pytest.raises(zg.ZignalError):
        zg.Image.from_numpy(np.zeros((8, 8, 4), np.uint8)).gaussian_blur(0.6)


# ---- the container tests of the reference's Python binding (bindings/python/tests/test_image.py) --------------------------------

def test_oracle_fill_and_set_border_known_answers(oracle):
    """test_image.py:110-142 (set_border) on the oracle's restatement of Image.fill / Image.setBorder (image.zig:191-230)."""
    img = np.zeros((4, 4, 3), np.uint8)
    oracle.fill(img, (10, 20, 30))
    assert (img == (10, 20, 30)).all()
    oracle.set_border(img, (1, 1, 3, 3))
    for corner in ((0, 0), (0, 3), (3, 0), (3, 3)):
        assert (img[corner] == 0).all()
    assert (img[1:3, 1:3] == (10, 20, 30)).all() and (img[0] == 0).all() and (img[:, 0] == 0).all()
    oracle.fill(img, (10, 20, 30))
    oracle.set_border(img, (1, 1, 3, 3), (255, 0, 0))
    assert tuple(img[0, 0]) == (255, 0, 0) and tuple(img[1, 1]) == (10, 20, 30)
    small = np.full((3, 3, 3), (7, 8, 9), np.uint8)  # a rectangle that misses the image fills all of it
    oracle.set_border(small, (10, 10, 20, 20))
    assert not small.any()
    clipped = np.full((5, 6), 9, np.uint8)           # a rectangle reaching past the image is clipped to it
    oracle.set_border(clipped, (2, 1, 60, 50))
    assert (clipped[1:, 2:] == 9).all() and not clipped[0].any() and not clipped[:, :2].any()


def test_mirror_container_api_shapes():
    """test_image.py:10-14, :96-108, :144-152, :154-166 on the host flavour (no GPU needed for the plumbing)."""
    img = zg.Image.from_numpy(np.zeros((3, 4, 4), np.uint8))
    assert (img.rows, img.cols) == (3, 4) and img.is_contiguous() is True
    v = zg.Image(np.zeros((4, 4, 4), np.uint8)).view((1, 1, 3, 3))
    assert (v.rows, v.cols) == (2, 2)
    assert zg.Image(np.zeros((5, 7), np.uint8)).get_rectangle() == (0, 0, 7, 5)
    arr = np.full((2, 3, 3), (1, 2, 3), np.uint8)
    assert np.array_equal(zg.Image.from_numpy(zg.Image.from_numpy(arr).to_numpy()).data, arr)
    with pytest.raises(ValueError):
        zg.Image.from_numpy(np.zeros((2, 3, 2), np.uint8))
    with pytest.raises(TypeError):
        zg.Image(np.zeros((3, 3, 3), np.uint8)).set_border(None)          # :139-142
    with pytest.raises(ValueError):
        zg.Image(np.zeros((3, 3, 3), np.uint8))._pixel_value((1, 2))
    # where this mirror is deliberately wider than the reference's Python binding: (rows, cols) arrays and float32 pixels are
    # Image(u8) / Image(f32) / Image(Rgb(f32)) / Image(Rgba(f32)) of the Zig API, which the path computes on
    assert zg.Image.from_numpy(np.zeros((2, 3), np.uint8)).pixel == 0 and zg.Image.from_numpy(np.zeros((2, 3, 3), np.float32)).pixel == 4


@pytest.mark.gpu
def test_fill_and_set_border_parity(oracle):
    import torch
    from tests.util import ALL_TYPES, synth
    rng = np.random.default_rng(3)
    for kind in ALL_TYPES:
        for (h, w) in ((1, 1), (5, 7), (33, 300), (64, 1025)):
            host = synth(oracle, kind, 5, h, w)
            ch = 1 if host.ndim == 2 else host.shape[2]
            value = (rng.random(ch).astype(np.float32) if host.dtype == np.float32 else rng.integers(0, 256, ch).astype(np.uint8))
            value = value[0] if ch == 1 else value
            for rect in ((1, 1, w - 1, h - 1), (0, 0, w, h), (w // 3, h // 2, w + 50, h + 9), (w + 5, 0, w + 9, 4), (2, 2, 2, 9), (0, 0, 1, 1)):
                for val in (None, value):
                    dev = zg.Image(torch.from_numpy(host.copy()).cuda())
                    dev.set_border(rect, val)
                    want = oracle.set_border(host.copy(), rect, val)
                    torch.cuda.synchronize()
                    assert np.array_equal(dev.to_numpy().view(np.uint8), want.view(np.uint8)), (kind, h, w, rect)
            dev = zg.Image(torch.from_numpy(host.copy()).cuda())
            assert np.array_equal(dev.fill(value).to_numpy().view(np.uint8), oracle.fill(host.copy(), value).view(np.uint8))
            hv = zg.Image(host.copy())  # host flavour: zg_fill_host / zg_set_border_host
            hv.set_border((1, 0, w - 1, h), value)
            assert np.array_equal(hv.data.view(np.uint8), oracle.set_border(host.copy(), (1, 0, w - 1, h), value).view(np.uint8))
    # a view: only the view's pixels change
    big = zg.Image(torch.full((20, 30, 4), 7, dtype=torch.uint8, device="cuda"))
    big.view((5, 4, 25, 16)).set_border((2, 2, 10, 6), (1, 2, 3, 4))
    out = big.to_numpy()
    want = np.full((20, 30, 4), 7, np.uint8)
    oracle.set_border(want[4:16, 5:25], (2, 2, 10, 6), (1, 2, 3, 4))
    assert np.array_equal(out, want)
