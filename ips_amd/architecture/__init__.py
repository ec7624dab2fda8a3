from .ips_net import IPSNet
from .transformer import (MLP, MultiHeadCrossAttention, ScaledDotProductAttention, Transformer,
                          pos_enc_1d)

__all__ = ["IPSNet", "Transformer", "MultiHeadCrossAttention", "ScaledDotProductAttention", "MLP",
           "pos_enc_1d"]
