"""goai_nn_forward_static (include/goai_hip.h): the CNN forward for callers that cannot allocate.

It has no counterpart in the reference.  Same kernels as goai_nn_forward on the same context, but
the activation buffers between them are the library's file-scope device array, so its outputs must
equal goai_nn_forward's bit for bit.  Two calls in a row through the C ABI, the second with a
smaller input (the array keeps the first call's contents), for every trunk kernel family; the
caller's output tensors are allocated before the calls, and sizes beyond the array are rejected.
"""
import numpy as np
import pytest
import torch

from helpers import random_positions
from oracle import nn_ref

pytestmark = pytest.mark.gpu


def _ctx(kind, size, layers, max_batch, seed=3, dtype=0):
    from go_ai.engine import Context
    c = Context(size)
    c.nn_init(Context.NET_AC if kind == "ac" else Context.NET_VAL, layers, max_batch, dtype=dtype)
    c.load_weights(nn_ref.make_state_dict(kind, size, seed, layers))
    return c


@pytest.mark.parametrize("kind,size,layers,max_batch,n1,n2,dtype", [
    ("ac", 9, 3, 1024, 101, 37, 0),     # F(3x3,3x3) paired trunk in one launch; 101 boards = the limit at 9x9
    ("ac", 9, 3, 4096, 90, 5, 0),       # F(3x3,3x3), one chunk
    ("val", 9, 3, 64, 64, 1, 0),        # direct implicit GEMM, value head
    ("ac", 13, 3, 512, 48, 17, 0),      # F(2x2,3x3); 48 boards = the limit at 13x13
    ("ac", 19, 10, 512, 22, 3, 0),      # the deeper trunk; 22 boards = the limit at 19x19
    ("ac", 5, 3, 16, 40, 7, 0),         # slices of max_batch = 16 inside one call
    ("ac", 9, 3, 1024, 101, 37, 3),  # GOAI_DTYPE_F16X3 (what bench.py runs): F(3x3,3x3), heads' 1x1 in the trunk launch
    ("ac", 13, 3, 64, 48, 17, 3),    # ... F(2x2,3x3): the boards' bounds travel between launches
    ("val", 9, 3, 16, 40, 3, 3),     # ... slices, value head
])
def test_static_forward_equals_forward_twice(kind, size, layers, max_batch, n1, n2, dtype):
    c = _ctx(kind, size, layers, max_batch, dtype=dtype)
    A = size * size + 1
    states = torch.from_numpy(random_positions(size, n1, seed=size + n1, canonical=True, pass_prob=0.05)).to(c.device)
    for mode in (c.NN_RAW, c.NN_NUMPY_SHIM):
        for n in (n1, n2):                       # second call smaller: stale rows of the first stay in the array
            st = states[:n].contiguous()
            want_pi, want_val = c.nn_forward(st, mode=mode)
            pi = torch.full((n, A), float("nan"), dtype=torch.float32, device=c.device) if kind == "ac" else None
            val = torch.full((n, 1), float("nan"), dtype=torch.float32, device=c.device)
            c.nn_forward_static(st, pi, val, mode=mode)
            assert np.array_equal(val.cpu().numpy(), want_val.cpu().numpy()), (mode, n)
            if kind == "ac":
                assert np.array_equal(pi.cpu().numpy(), want_pi.cpu().numpy()), (mode, n)
    c.close()


def test_static_forward_rejects_sizes_beyond_its_array():
    from go_ai import _hip
    c = _ctx("ac", 9, 3, 1024)
    n = c.NN_STATIC_MAX_CELLS // 81 + 1
    st = torch.zeros((n, 6, 9, 9), dtype=torch.uint8, device=c.device)
    pi = torch.empty((n, 82), dtype=torch.float32, device=c.device)
    val = torch.empty((n, 1), dtype=torch.float32, device=c.device)
    with pytest.raises(_hip.GoaiError, match="GOAI_NN_STATIC_MAX_CELLS"):
        c.nn_forward_static(st, pi, val)
    c.nn_forward_static(st[:n - 1].contiguous(), pi[:n - 1], val[:n - 1])      # the limit itself is fine
    torch.cuda.synchronize()
    assert torch.isfinite(val[:n - 1]).all()
    c.close()


@pytest.mark.parametrize("kind,size,layers,max_batch,n,dtype", [("ac", 9, 3, 1024, 101, 3), ("ac", 9, 3, 1024, 64, 0),
                                                                 ("ac", 13, 3, 64, 48, 3), ("val", 9, 3, 16, 40, 3)])
def test_static_forward_allocates_no_device_memory(kind, size, layers, max_batch, n, dtype):
    """The property the entry point exists for: with the caller's inputs and outputs in place, a call leaves the
    device's free memory where it was (hipMemGetInfo through torch.cuda.mem_get_info -- the driver's own figure, which
    sees hipMalloc from any library, torch's caching allocator included).  Checked after one warm-up call, over several
    calls in both output modes.  (HIP graph capture of the call is NOT tested here: the header only promises that the
    call allocates nothing.)"""
    c = _ctx(kind, size, layers, max_batch, dtype=dtype)
    A = size * size + 1
    st = torch.from_numpy(random_positions(size, n, seed=1, canonical=True, pass_prob=0.05)).to(c.device)
    pi = torch.empty((n, A), dtype=torch.float32, device=c.device) if kind == "ac" else None
    val = torch.empty((n, 1), dtype=torch.float32, device=c.device)
    c.nn_forward_static(st, pi, val, mode=c.NN_RAW)           # warm-up: code objects loaded, lazy state set up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(c.device)
    reserved0 = torch.cuda.memory_reserved(c.device)
    for mode in (c.NN_RAW, c.NN_NUMPY_SHIM, c.NN_RAW):
        c.nn_forward_static(st, pi, val, mode=mode)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(c.device)
    assert free1 == free0, (free0, free1)
    assert torch.cuda.memory_reserved(c.device) == reserved0
    c.close()
