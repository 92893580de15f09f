"""ValueNet (/root/reference/go_ai/models/val_net.py:8-27) on the HIP forward."""
from . import RLNet


class ValueNet(RLNet):
    kind = "val"
