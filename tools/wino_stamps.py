"""Diagnostic (needs a -DGOAI_CONV_STAMPS build named by GOAI_HIP_LIB, GOAI_CONV_ALGO=wino): phases of
the Winograd conv workgroups from s_memrealtime stamps (100 MHz)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd")); sys.path.insert(0, ROOT)
from go_ai import _hip
from go_ai.engine import Context
from go_ai.models import random_state_dict
c = Context(9); c.nn_init(0, 3, 1024); c.load_weights(random_state_dict("ac", 9, 0))
st = torch.from_numpy((np.random.default_rng(0).random((1024, 6, 9, 9)) < 0.3).astype(np.uint8)).cuda()
for _ in range(3): c.nn_forward(st)
torch.cuda.synchronize()
n = int(os.environ.get("NWG", "256"))
buf = (ctypes.c_ulonglong * (8 * n))()
getattr(_hip.lib, os.environ.get("STAMP_SYM", "goai_debug_wino3_stamps" if os.environ.get("GOAI_CONV_ALGO") == "wino3" else "goai_debug_wino_stamps"))(buf, 8 * n)
a = np.array(buf, dtype=np.int64).reshape(n, 8)
t0 = a[:, 0].min()
t = (a[:, :6] - t0) / 100.0
names = ["start", "prologue", "chunk 0", "chunks 1-2", "chunk 3", "epilogue"]
pr = lambda name, x: print(f"{name}: min {x.min():.1f} p50 {np.median(x):.1f} p90 {np.percentile(x, 90):.1f} max {x.max():.1f}")
pr(names[0], t[:, 0])
for i in range(1, 6): pr(names[i], t[:, i] - t[:, i - 1])
pr("end", t[:, 5])
cyc = (a[:, 7] - a[:, 6]).astype(float); us = (a[:, 4] - a[:, 1]) / 100.0
print(f"main loop: {np.median(cyc):.0f} shader cycles in {np.median(us):.1f} us -> in-kernel clock {np.median(cyc / us) / 1e3:.3f} GHz; {np.median(cyc) / 64:.0f} cycles per 1/64 of the loop (MFMA pipe needs 3584 for F(2x2,3x3), 1536 for F(3x3,3x3))")
