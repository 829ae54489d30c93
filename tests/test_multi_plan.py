"""zg_multi's shard / piece arithmetic without a GPU (zg_multi_piece_range is host only): the cut a one-thread C++ / Zig host makes
(zignal_amd/csrc/zg_multi.cpp) against the cut the one-process-per-GPU route makes (zignal_amd/sharding.py), and the properties both need —
pieces disjoint, in order, covering the batch, shard and piece sizes within one frame of each other (contiguous blocks of frames per
GPU, no halo). world 1 .. 8, ragged batches, batches shorter than the world, every piece count the context accepts."""
import pytest

import zignal_amd as zg
from zignal_amd import _lib as L
from zignal_amd import sharding
from zignal_amd.pipeline import piece_range


@pytest.mark.parametrize("world", (1, 2, 3, 4, 7, 8))
def test_pieces_partition_the_batch_like_sharding_py(world):
    for n in (0, 1, 2, 5, 8, 31, 128, 1024, 1027):
        for chunks in (1, 3, 4, 8):
            seen = []
            for dev in range(world):
                sb, se = sharding.shard_range(n, dev, world)
                cuts = sharding.chunk_ranges(se - sb, chunks)
                for c in range(8):
                    b, e = piece_range(n, world, chunks, dev, c) if c < chunks else (sb, sb)
                    if c < len(cuts):
                        assert (b, e) == (sb + cuts[c][0], sb + cuts[c][1]), (n, world, chunks, dev, c)
                        assert e > b
                        seen.append((b, e))
                    else:
                        assert b == e, "pieces past a shard's last are empty"
            flat = [f for b, e in seen for f in range(b, e)]
            assert flat == list(range(n)), (n, world, chunks)  # in order, disjoint, covering
            sizes = [se - sb for sb, se in (sharding.shard_range(n, d, world) for d in range(world))]
            assert max(sizes) - min(sizes) <= 1


def test_piece_range_rejects_bad_arguments():
    lib = zg.lib()
    import ctypes as C
    b, e = C.c_uint32(), C.c_uint32()
    for args in ((10, 0, 4, 0, 0), (10, 2, 4, 2, 0), (10, 2, 0, 0, 0), (10, 2, 9, 0, 0), (10, 2, 4, -1, 0), (10, 2, 4, 0, -1)):
        assert lib.zg_multi_piece_range(*args, C.byref(b), C.byref(e)) == L.ERR_INVALID_ARGUMENT, args
    assert lib.zg_multi_piece_range(10, 2, 4, 0, 0, None, None) == L.ERR_INVALID_ARGUMENT


def test_binding_and_library_agree_on_zg_step():
    import ctypes as C
    assert zg.lib().zg_sizeof_step() == C.sizeof(L.ZgStep)
