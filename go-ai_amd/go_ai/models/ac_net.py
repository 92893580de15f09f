"""ActorCriticNet (/root/reference/go_ai/models/ac_net.py:8-64) on the HIP forward."""
from . import RLNet


class ActorCriticNet(RLNet):
    kind = "ac"

    def __init__(self, size, **kw):
        super().__init__(size, **kw)
        self.action_size = size * size + 1
