"""Diagnostic (needs a -DGOAI_CONV_STAMPS build named by GOAI_HIP_LIB): phases of the bf16x3 conv workgroups."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd")); sys.path.insert(0, ROOT)
from go_ai import _hip
from go_ai.engine import Context
from go_ai.models import random_state_dict
c = Context(9); c.nn_init(0, 3, 1024, dtype=1); c.load_weights(random_state_dict("ac", 9, 0))
st = torch.from_numpy((np.random.default_rng(0).random((1024, 6, 9, 9)) < 0.3).astype(np.uint8)).cuda()
for _ in range(3): c.nn_forward(st)
torch.cuda.synchronize()
n = int(os.environ.get("NWG", "432"))
buf = (ctypes.c_ulonglong * (8 * n))()
_hip.lib.goai_debug_bf_stamps(buf, 8 * n)
a = np.array(buf, dtype=np.int64).reshape(n, 8)
t0 = a[:, 0].min()
s, l0, l1, e = [(a[:, i] - t0) / 100.0 for i in range(4)]
pr = lambda name, x: print(f"{name}: min {x.min():.1f} p50 {np.median(x):.1f} p90 {np.percentile(x, 90):.1f} max {x.max():.1f}")
pr("start", s); pr("prologue", l0 - s); pr("loop", l1 - l0); pr("epilogue", e - l1); pr("end", e)
for lo in range(0, n, 128):
    m = slice(lo, min(n, lo + 128))
    print(f"blocks {lo}-{lo+127}: start {s[m].mean():.1f} loop {np.mean(l1[m]-l0[m]):.1f} end {e[m].mean():.1f}")
