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
