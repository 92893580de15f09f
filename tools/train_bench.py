"""ms per goai_train_step (GPU) next to the torch-CPU oracle's train_step, same batches (run on the GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "go-ai_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from go_ai.engine import Context  # noqa: E402
from oracle import nn_ref, nn_train_ref  # noqa: E402

torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "16")))
for kind, size, n in (("ac", 9, 32), ("ac", 9, 256), ("val", 9, 32), ("ac", 19, 32)):
    rng = np.random.default_rng(0)
    sd = nn_ref.make_state_dict(kind, size, 1)
    c = Context(size, 0)
    c.train_init(0 if kind == "ac" else 1, 3, n)
    c.train_load(sd)
    states = (rng.random((n, 6, size, size)) < 0.3).astype(np.uint8)
    nxt = (rng.random((n, 6, size, size)) < 0.3).astype(np.uint8)
    wins = rng.integers(-1, 2, n).astype(np.float32)
    pi = rng.random((n, size * size + 1)).astype(np.float32)
    dev = [torch.from_numpy(a).cuda() for a in (states, nxt, wins, pi)]
    for _ in range(3):
        c.train_step(*dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 30
    for _ in range(steps):
        c.train_step(*dev)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / steps * 1e3
    adam = nn_train_ref.AdamState(sd)
    f = [states.astype(np.float32), nxt.astype(np.float32), wins, pi]
    nn_train_ref.train_step(sd, kind, adam, *f)
    t0 = time.perf_counter()
    cpu_steps = 3
    for _ in range(cpu_steps):
        nn_train_ref.train_step(sd, kind, adam, *f)
    cpu_ms = (time.perf_counter() - t0) / cpu_steps * 1e3
    print(f"{kind} {size}x{size} batch {n}: GPU {gpu_ms:.2f} ms/step ({n / gpu_ms * 1e3:.0f} positions/s), "
          f"torch CPU ({torch.get_num_threads()} threads) {cpu_ms:.1f} ms/step", flush=True)
    c.close()
