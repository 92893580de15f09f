"""Throughput of the stateless Go-rule kernels (goai_transition / goai_children / goai_score) on large
batches: boards/s and ALGORITHMIC HBM bytes/s (6P bytes in + 6P bytes out per
transition, 6P in + A*6P out per children call, 6P in per score) against the 8 TB/s HBM peak."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from go_ai.engine import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", type=int, nargs="+", default=[9, 19])
ap.add_argument("--batch", type=int, default=1 << 18)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--moves", type=int, default=-1, help="moves played from the empty board to make the positions (default: P / 2)")
args = ap.parse_args()


def positions(c, size, n, moves):
    """n mid-game positions made on the GPU itself: random legal non-pass moves from the empty board."""
    st = torch.zeros((n, 6, size, size), dtype=torch.uint8, device="cuda")
    c.mcts_init(n, 0)
    for i in range(moves):
        c.mcts_reset(st)
        acts = c.sample_actions(c.uniform_policy(allow_pass=False), 99, i)
        nxt, status = c.transition(st, acts, canonical=True)
        st = torch.where((status == 0)[:, None, None, None], nxt, st)
    return st


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for size in args.sizes:
    P, A = size * size, size * size + 1
    n = args.batch
    c = Context(size)
    st = positions(c, size, n, moves=P // 2 if args.moves < 0 else args.moves)
    stones = float((st[:, 0] | st[:, 1]).sum()) / n
    acts = c.sample_actions(c.uniform_policy(allow_pass=False), 7, 0)
    out, status = torch.empty_like(st), torch.empty(n, dtype=torch.int32, device="cuda")
    from go_ai import _hip
    t = timed(lambda: _hip.check(_hip.lib.goai_transition(c._h, _hip.ptr(st), _hip.ptr(acts), 1, _hip.ptr(out),
                                                          _hip.ptr(status), n)), args.iters)
    print(f"{size}x{size} transition: {n} boards ({stones:.0f} stones avg) {t * 1e3:.2f} ms  {n / t / 1e6:.1f} M boards/s  "
          f"{n * 12 * P / t / 1e9:.0f} GB/s algorithmic ({n * 12 * P / t / 8e12:.3f} of HBM peak)")
    b, w, e, win = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(4)]
    t = timed(lambda: _hip.check(_hip.lib.goai_score(c._h, _hip.ptr(st), _hip.ptr(b), _hip.ptr(w), _hip.ptr(e),
                                                     _hip.ptr(win), n)), args.iters)
    print(f"{size}x{size} score:      {n} boards {t * 1e3:.2f} ms  {n / t / 1e6:.1f} M boards/s  "
          f"{n * 6 * P / t / 1e9:.0f} GB/s algorithmic")
    nc = max(1, n // 64)
    kids = torch.empty((nc, A, 6, size, size), dtype=torch.uint8, device="cuda")
    valid = torch.empty((nc, A), dtype=torch.uint8, device="cuda")
    win8 = torch.empty((nc, A), dtype=torch.int8, device="cuda")
    t = timed(lambda: _hip.check(_hip.lib.goai_children(c._h, _hip.ptr(st), 1, _hip.ptr(kids), _hip.ptr(valid),
                                                        _hip.ptr(win8), nc)), args.iters)
    print(f"{size}x{size} children:   {nc} boards x {A} actions {t * 1e3:.2f} ms  {nc * A / t / 1e6:.1f} M transitions/s  "
          f"{nc * (A + 1) * 6 * P / t / 1e9:.0f} GB/s algorithmic ({nc * (A + 1) * 6 * P / t / 8e12:.3f} of HBM peak)")
    c.close()
