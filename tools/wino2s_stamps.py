"""Diagnostic (needs a -DGOAI_CONV_STAMPS build of conv_wino2s_kernels.hip in the library named by GOAI_HIP_LIB, see
tools/wino2s_stamps.sh): phases of one F(2x2,3x3) convolution launch (the forward's last full one), from s_memrealtime stamps
(100 MHz).  S = board size, NB = boards, LAYERS = blocks."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd")); sys.path.insert(0, ROOT)
from go_ai import _hip
from go_ai.engine import Context
from go_ai.models import random_state_dict
S = int(os.environ.get("S", "13")); NB = int(os.environ.get("NB", "8192")); LAYERS = int(os.environ.get("LAYERS", "3"))
DT = {"f16x3": Context.DTYPE_F16X3, "bf16x6": Context.DTYPE_BF16X6, "f16": Context.DTYPE_F16}[os.environ.get("DTYPE", "f16x3")]
c = Context(S); c.nn_init(0, LAYERS, NB, dtype=DT); c.load_weights(random_state_dict("ac", S, 0, layers=LAYERS) if LAYERS != 3 else random_state_dict("ac", S, 0))
st = torch.from_numpy((np.random.default_rng(0).random((NB, 6, S, S)) < 0.3).astype(np.uint8)).cuda()
for _ in range(2): c.nn_forward(st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): c.nn_forward(st)
e1.record(); torch.cuda.synchronize()
print(f"forward of {NB} boards {S}x{S}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us")
T2 = ((S + 1) // 2) ** 2; nbw = 112 // T2; n = min((NB + nbw - 1) // nbw, 8192)
buf = (ctypes.c_ulonglong * (40 * n))()
_hip.lib.goai_debug_wino2s_stamps(buf, 40 * n)
a = np.array(buf, dtype=np.int64).reshape(n, 40)
t = (a[:, :35] - a[:, :1]) / 100.0
med = np.median(t, axis=0)
print(f"prologue (zero fill, first image staged) {med[1]:7.2f} us")
tr = [med[2 + 2 * r] - med[1 + 2 * r] for r in range(16)]
stg = [med[3 + 2 * r] - med[2 + 2 * r] for r in range(16)]
for r in range(16):
    print(f"row chunk={r // 4} p={r % 4}: transform {tr[r]:6.2f} us   4 stages {stg[r]:6.2f} us")
print(f"epilogue            {med[34] - med[33]:7.2f} us")
e = (a[:, [37, 38, 34]] - a[:, 33:34]) / 100.0
print(f"  epilogue: first barrier passed {np.median(e[:, 0]):.2f}, wave 0's last store issued {np.median(e[:, 1]):.2f}, end {np.median(e[:, 2]):.2f} us after wave 0's last stage")
print(f"workgroup           {med[34]:7.2f} us   (transforms {sum(tr):.1f}, stages {sum(stg):.1f})")
cyc = np.median((a[:, 36] - a[:, 35]).astype(float)); us = np.median((a[:, 34] - a[:, 0]) / 100.0)
print(f"in-kernel clock {cyc / us / 1e3:.3f} GHz; a stage = {sum(stg) / 64 * cyc / us:.0f} shader cycles")
print(f"workgroup time over workgroups: min {t[:, 34].min():.1f} p10 {np.percentile(t[:, 34], 10):.1f} p90 {np.percentile(t[:, 34], 90):.1f} max {t[:, 34].max():.1f}")
span = (a[:, 34].max() - a[:, 0].min()) / 100.0
print(f"launch span (first top to last end) {span:.1f} us; workgroups {n}; rounds {n / 256:.1f}")
ep = (a[:, 34] - a[:, 33]) / 100.0
print(f"epilogue over workgroups: p10 {np.percentile(ep, 10):.1f} median {np.median(ep):.1f} p90 {np.percentile(ep, 90):.1f}")
