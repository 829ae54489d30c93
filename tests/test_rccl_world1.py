"""The RCCL path executed for real on the one GPU a test box has: `init_process_group("nccl")` at world size 1, a barrier
(ncclAllReduce), and the scatter -> [blur, resize] -> gather fan-out of BASELINE config 5 with the root's own shard sent
through the communicator (`loopback`: a grouped ncclSend / ncclRecv to itself) instead of copied. world > 1 needs more
GPUs than a test box has; the gloo tests cover its ownership logic, this one covers that the nccl backend initialises,
moves frames and interoperates with the library's stream use."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ZG_ROOT"])
import zignal_amd as zg
from zignal_amd import sharding
from oracle import pyoracle as oracle  # checker

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
lib = zg.lib()
assert lib.zg_init(0) == 0
dist.barrier()                                        # ncclAllReduce on the one-rank communicator
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.5
n, rows, cols = 4, 270, 480
host = oracle.synth_u8(17, (n, rows, cols, 4))
batch = torch.from_numpy(host).to(dev)
mine = sharding.scatter_frames(batch, n, (rows, cols, 4), torch.uint8, dev, loopback=True)   # ncclSend + ncclRecv to self
assert torch.equal(mine, batch)
out = torch.empty((n, rows // 2, cols // 2, 4), dtype=torch.uint8, device=dev)
m = zg.Interpolation.bilinear._c()
rc = lib.zg_batch_blur_resize(C.c_void_p(mine.data_ptr()), n, rows, cols, 3, C.c_float(0.6), C.c_void_p(out.data_ptr()),
                              rows // 2, cols // 2, C.byref(m), C.c_void_p(torch.cuda.current_stream().cuda_stream))
assert rc == 0, lib.zg_last_error()
back = sharding.gather_frames(out, n, loopback=True)
torch.cuda.synchronize()
got = back.cpu().numpy()
bil = oracle.method(oracle.BILINEAR)
for f in range(n):
    want = oracle.resize(oracle.gaussian_blur(host[f], 0.6), (rows // 2, cols // 2), bil)
    assert np.array_equal(got[f], want), f"frame {f} differs from the oracle after the RCCL round trip"
assert sharding.max_over_ranks(1.25, dev) == 1.25
assert sharding.count_ranks(dev) == 1 and sharding.per_rank(2.5, dev) == [2.5]

# the chunked form of the same exchange (what bench.py --scatter-gather times): three pieces, results back on a second communicator
second = dist.new_group(backend="nccl")
def blur_resize(src, dst):
    rc = lib.zg_batch_blur_resize(C.c_void_p(src.data_ptr()), int(src.shape[0]), rows, cols, 3, C.c_float(0.6), C.c_void_p(dst.data_ptr()),
                                  rows // 2, cols // 2, C.byref(m), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.zg_last_error()
for chunks in (3, 1, 8):
    piped = sharding.scatter_compute_gather(batch, n, (rows, cols, 4), (rows // 2, cols // 2, 4), torch.uint8, dev, blur_resize, chunks=chunks,
                                            loopback=True, gather_group=second)
    torch.cuda.synchronize()
    assert np.array_equal(piped.cpu().numpy(), got), f"pipelined exchange with {chunks} pieces differs"
dist.destroy_process_group()
print("rccl world1 ok")
'''


@pytest.mark.gpu
def test_nccl_backend_scatter_blur_resize_gather_at_world_size_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               ZG_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "rccl world1 ok" in out.stdout, out.stdout + out.stderr
