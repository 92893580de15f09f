"""A few goai_train_step calls for rocprofv3 (actor-critic, 9x9; batch from argv)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "go-ai_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from go_ai.engine import Context  # noqa: E402
from go_ai.models import random_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 9
rng = np.random.default_rng(0)
c = Context(size, 0)
c.train_init(0, 3, n)
c.train_load(random_state_dict("ac", size, 1))
dev = [torch.from_numpy(a).cuda() for a in ((rng.random((n, 6, size, size)) < 0.3).astype(np.uint8),
                                            (rng.random((n, 6, size, size)) < 0.3).astype(np.uint8),
                                            rng.integers(-1, 2, n).astype(np.float32),
                                            rng.random((n, size * size + 1)).astype(np.float32))]
for _ in range(20):
    c.train_step(*dev)
torch.cuda.synchronize()
c.close()
