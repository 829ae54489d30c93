"""Shared helpers for the GPU parity tests."""
import numpy as np

SHAPES = {"u8": (), "f32": (), "rgb_u8": (3,), "rgba_u8": (4,), "rgb_f32": (3,), "rgba_f32": (4,)}
IS_FLOAT = {"u8": False, "f32": True, "rgb_u8": False, "rgba_u8": False, "rgb_f32": True, "rgba_f32": True}
ALL_TYPES = tuple(SHAPES)


def synth(oracle, kind: str, seed: int, rows: int, cols: int) -> np.ndarray:
    shape = (rows, cols) + SHAPES[kind]
    return oracle.synth_f32(seed, shape) if IS_FLOAT[kind] else oracle.synth_u8(seed, shape)


def assert_bits_equal(got: np.ndarray, want: np.ndarray, what: str = ""):
    assert got.shape == want.shape and got.dtype == want.dtype, f"{what}: {got.shape}/{got.dtype} vs {want.shape}/{want.dtype}"
    g, w = np.ascontiguousarray(got), np.ascontiguousarray(want)
    if g.dtype == np.float32:
        g, w = g.view(np.uint32), w.view(np.uint32)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {g.size} elements differ; first at {first}: "
                             f"got {got[first]!r} want {want[first]!r}")


def ulp_diff(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2147483648) - ai, ai)
    bi = np.where(bi < 0, np.int64(-2147483648) - bi, bi)
    return np.abs(ai - bi)
