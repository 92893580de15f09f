"""Diagnostic (needs a -DGOAI_CONV_STAMPS build of conv_wino3h_kernels.hip in the library named by GOAI_HIP_LIB,
see tools/wino3h_stamps.sh): phases of the second convolution of the split F(3x3,3x3) trunk launch (DTYPE=f16x3 | bf16x6), from
s_memrealtime stamps (100 MHz)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd")); sys.path.insert(0, ROOT)
from go_ai import _hip
from go_ai.engine import Context
from go_ai.models import random_state_dict
NB = int(os.environ.get("NB", "1024"))
F16 = os.environ.get("DTYPE", "f16x3") == "f16x3"
c = Context(9); c.nn_init(0, 3, 1024, dtype=Context.DTYPE_F16X3 if F16 else Context.DTYPE_BF16X6); c.load_weights(random_state_dict("ac", 9, 0))
assert c.nn_conv_algo() == (Context.CONV_WINOGRAD_F3_F16X3 if F16 else Context.CONV_WINOGRAD_F3_BF16X6)
st = torch.from_numpy((np.random.default_rng(0).random((NB, 6, 9, 9)) < 0.3).astype(np.uint8)).cuda()
for _ in range(3): c.nn_forward(st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): c.nn_forward(st)
e1.record(); torch.cuda.synchronize()
print(f"forward of {NB} boards: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
n = (NB + 3) // 4
buf = (ctypes.c_ulonglong * (32 * n))()
_hip.lib.goai_debug_wino3h_stamps(buf, 32 * n)
a = np.array(buf, dtype=np.int64).reshape(n, 32)
t = (a[:, :23] - a[:, :1]) / 100.0                       # us since the layer's top barrier, per workgroup
med = np.median(t, axis=0)
print(f"image staged        {med[1]:7.2f} us")
tr = [med[2 + 2 * r] - (med[1] if r == 0 else med[1 + 2 * r]) for r in range(10)]
stg = [med[3 + 2 * r] - med[2 + 2 * r] for r in range(10)]
for r in range(10):
    print(f"row cp={r // 5} p={r % 5}: transform {tr[r]:6.2f} us   5 stages {stg[r]:6.2f} us")
print(f"epilogue            {med[22] - med[21]:7.2f} us")
print(f"layer               {med[22]:7.2f} us   (transforms {sum(tr):.1f}, stages {sum(stg):.1f})")
cyc = np.median((a[:, 24] - a[:, 23]).astype(float)); us = np.median((a[:, 22] - a[:, 0]) / 100.0)
print(f"in-kernel clock {cyc / us / 1e3:.3f} GHz; a stage = {sum(stg) / 50 * cyc / us:.0f} shader cycles (its {18 if F16 else 36} MFMAs x 2 waves need {576 if F16 else 1152} per SIMD)")
print(f"spread over workgroups of the layer time: min {t[:, 22].min():.1f} max {t[:, 22].max():.1f}")
