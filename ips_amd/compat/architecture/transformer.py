from ips_amd.architecture.transformer import (MLP, MultiHeadCrossAttention,  # noqa: F401
                                              ScaledDotProductAttention, Transformer, pos_enc_1d)
