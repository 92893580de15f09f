"""Diagnostic (needs a -DGOAI_PF_STAMPS build of mcts_kernels.hip named by GOAI_HIP_LIB): phases of k_post_forward over the
workgroups of its last launch in a C2-shaped search, from s_memrealtime stamps (100 MHz)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd")); sys.path.insert(0, ROOT)
from go_ai import _hip
from go_ai.game import BatchedSelfPlay
from go_ai.engine import Context
sp = BatchedSelfPlay(9, 1024, 81, seed=1234, dtype=Context.DTYPE_F16X3, refill=True)
sp.play(3)
sp.sync()
n = 512
buf = (ctypes.c_ulonglong * (8 * n))()
_hip.lib.goai_debug_pf_stamps(buf, 8 * n)
a = np.array(buf, dtype=np.int64).reshape(n, 8)
t = (a[:, :6] - a[:, :1]) / 100.0
med = np.median(t, axis=0)
names = ["heads+barrier", "backup", "select", "barrier", "conv0"]
for i in range(1, 6):
    print(f"{names[i - 1]:14s} {med[i] - med[i - 1]:6.2f} us (ends at {med[i]:6.2f})")
print(f"workgroup total: median {med[5]:.2f} us, max {t[:, 5].max():.2f} us; launch span {(a[:, 5].max() - a[:, 0].min()) / 100.0:.2f} us")
