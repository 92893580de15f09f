"""Diagnostic (needs a -DGOAI_CONV_STAMPS build in /tmp/libgoai_stamps.so): per-workgroup start / end times of one conv3x3 launch."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-ai_amd")); sys.path.insert(0, ROOT)
from go_ai import _hip
from go_ai.engine import Context
from go_ai.models import random_state_dict
c = Context(9); c.nn_init(0, 3, 1024); c.load_weights(random_state_dict("ac", 9, 0))
st = torch.from_numpy((np.random.default_rng(0).random((int(os.environ.get("BATCH", "1024")), 6, 9, 9)) < 0.3).astype(np.uint8)).cuda()
for _ in range(3): c.nn_forward(st)
torch.cuda.synchronize()
n = int(os.environ.get("NWG", "741"))
buf = (ctypes.c_ulonglong * (8 * n))()
_hip.lib.goai_debug_conv_stamps(buf, 8 * n)
a = np.array(buf, dtype=np.int64).reshape(n, 8)
t0 = a[:, 0].min()
start, loop_end, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0   # 100 MHz -> us
print("WGs", n, "start us: min %.1f p50 %.1f p90 %.1f max %.1f" % (start.min(), np.median(start), np.percentile(start, 90), start.max()))
print("loop_end us: min %.1f p50 %.1f max %.1f" % (loop_end.min(), np.median(loop_end), loop_end.max()))
print("end us: min %.1f p50 %.1f max %.1f" % (end.min(), np.median(end), end.max()))
print("lifetime us: p50 %.1f max %.1f ; epilogue p50 %.1f max %.1f" % (np.median(end - start), (end - start).max(), np.median(end - loop_end), (end - loop_end).max()))
late = start > 20
print("late starters:", late.sum(), "xcc histogram of late:", np.bincount(a[late, 3].astype(int), minlength=8), "all:", np.bincount(a[:, 3].astype(int), minlength=8))
print("loop_end percentiles:", np.percentile(loop_end, [5, 25, 50, 75, 90, 95, 99]).round(1))
for x in range(8):
    m = a[:, 3] == x
    print("xcc", x, "n", m.sum(), "loop_end mean %.1f min %.1f max %.1f" % (loop_end[m].mean(), loop_end[m].min(), loop_end[m].max()))
idx = np.arange(n)
for lo in range(0, n, 96):
    m = (idx >= lo) & (idx < lo + 96)
    print("blocks %3d-%3d loop_end mean %.1f" % (lo, lo + 95, loop_end[m].mean()), end="; ")
print()
clk = (a[:, 5] - a[:, 4]) / ((a[:, 1] - a[:, 0]) / 100.0)   # shader cycles per us = MHz
print("in-kernel shader clock MHz: p5 %.0f p50 %.0f p95 %.0f" % tuple(np.percentile(clk, [5, 50, 95])))
