"""Quick per-kernel timing of the CNN forward (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd"))
sys.path.insert(0, ROOT)
from go_ai.engine import Context  # noqa: E402
from go_ai.models import random_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=9)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--layers", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", choices=["f32", "bf16x3"], default="f32")
args = ap.parse_args()

c = Context(args.size)
c.nn_init(Context.NET_AC, args.layers, args.batch, dtype=Context.DTYPE_BF16X3 if args.dtype == "bf16x3" else Context.DTYPE_F32)
c.load_weights(random_state_dict("ac", args.size, seed=0, layers=args.layers))
rng = np.random.default_rng(0)
states = torch.from_numpy((rng.random((args.batch, 6, args.size, args.size)) < 0.3).astype(np.uint8)).cuda()
for _ in range(3):
    c.nn_forward(states)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    c.nn_forward(states)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
P = args.size ** 2
flops = args.batch * (18 * P * (6 * 128 + 2 * args.layers * 128 * 128))
print(f"{args.dtype} size {args.size} batch {args.batch} layers {args.layers}: {ms:.3f} ms/forward, "
      f"{args.batch / ms * 1e3:.0f} boards/s, trunk {flops / ms / 1e9:.1f} TFLOP/s")
