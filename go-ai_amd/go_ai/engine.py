"""Thin object wrapper around a `goai_ctx` (include/goai_hip.h).

Holds one GPU context per (device, board size) and exposes the C-ABI entry points on torch
device tensors.  numpy inputs are accepted for convenience and copied to the GPU; results
come back as torch device tensors (callers `.cpu().numpy()` them where the reference's
surface promises numpy).
"""
import ctypes

import numpy as np
import torch

from . import _hip

NUM_CHNLS = 6


def _dev_u8_states(states, device):
    if isinstance(states, np.ndarray):
        if states.dtype != np.uint8:
            states = (states != 0).astype(np.uint8)
        states = torch.from_numpy(np.ascontiguousarray(states))
    if states.dtype != torch.uint8:
        states = (states != 0).to(torch.uint8)
    return states.to(device, non_blocking=True).contiguous()


class Context:
    def __init__(self, board_size, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("go_ai.engine needs a ROCm GPU; there is no CPU fallback")
        self.size = int(board_size)
        self.P = self.size * self.size
        self.A = self.P + 1
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = ctypes.c_void_p()
        _hip.check(_hip.lib.goai_ctx_create(ctypes.byref(h), self.device_index, self.size))
        self._h = h
        self.set_stream(torch.cuda.current_stream(self.device))

    def close(self):
        if getattr(self, "_h", None) is not None:
            _hip.lib.goai_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        self._stream = stream
        _hip.check(_hip.lib.goai_ctx_set_stream(self._h, ctypes.c_void_p(stream.cuda_stream)))

    def sync(self):
        _hip.check(_hip.lib.goai_ctx_sync(self._h))

    # ---- Go rules ------------------------------------------------------------------
    def transition(self, states, actions, canonical=True):
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        acts = torch.as_tensor(np.asarray(actions) if not torch.is_tensor(actions) else actions)
        acts = acts.to(self.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(st)
        status = torch.empty(n, dtype=torch.int32, device=self.device)
        _hip.check(_hip.lib.goai_transition(self._h, _hip.ptr(st), _hip.ptr(acts), int(bool(canonical)),
                                            _hip.ptr(out), _hip.ptr(status), n))
        return out, status

    def children(self, states, canonical=True):
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        kids = torch.empty((n, self.A, NUM_CHNLS, self.size, self.size), dtype=torch.uint8, device=self.device)
        valid = torch.empty((n, self.A), dtype=torch.uint8, device=self.device)
        win = torch.empty((n, self.A), dtype=torch.int8, device=self.device)
        _hip.check(_hip.lib.goai_children(self._h, _hip.ptr(st), int(bool(canonical)), _hip.ptr(kids),
                                          _hip.ptr(valid), _hip.ptr(win), n))
        return kids, valid, win

    def score(self, states):
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        outs = [torch.empty(n, dtype=torch.int32, device=self.device) for _ in range(4)]
        _hip.check(_hip.lib.goai_score(self._h, _hip.ptr(st), *[_hip.ptr(o) for o in outs], n))
        return tuple(outs)   # black_area, white_area, ended, winner
