"""Diagnostic (needs a -DGOAI_PF_STAMPS build named by GOAI_HIP_LIB, tools/pf_stamps.sh): phases of one step (the sixth) of
k_search_steps over the workgroups of its last launch in a C2-shaped search, from s_memrealtime stamps (100 MHz)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd")); sys.path.insert(0, ROOT)
os.environ["GOAI_STEPS_KERNEL"] = "1"
from go_ai import _hip
from go_ai.game import BatchedSelfPlay
from go_ai.engine import Context
sp = BatchedSelfPlay(9, 1024, 81, seed=1234, dtype=Context.DTYPE_F16X3, refill=True)
sp.play(3)
sp.sync()
n = 256
buf = (ctypes.c_ulonglong * (8 * n))()
_hip.lib.goai_debug_pf_stamps(buf, 8 * n)
a = np.array(buf, dtype=np.int64).reshape(n, 8)
t = (a[:, :5] - a[:, :1]) / 100.0
med = np.median(t, axis=0)
names = ["trunk + barrier", "heads", "backup + select", "conv0"]
for i in range(1, 5):
    print(f"{names[i - 1]:16s} {med[i] - med[i - 1]:7.2f} us (ends at {med[i]:7.2f})")
print(f"step: median {med[4]:.2f} us, min {t[:, 4].min():.2f} max {t[:, 4].max():.2f}; starts spread over {(a[:, 0].max() - a[:, 0].min()) / 100.0:.1f} us")
