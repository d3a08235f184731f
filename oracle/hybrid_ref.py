"""CPU oracle for the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product package never does; its forward fails
loudly when the HIP library is missing.

This is a stock ``torch.nn`` fp32 restatement of the reference's algorithm for
the hot path.  Every stage cites the reference file:line it
follows (paths relative to the reference repository root):

* conv stage      -- ``UNet._block`` first triple, ``UNet.py:58-60`` (Conv3x3
                     pad 1 bias=False -> BatchNorm2d -> ReLU) + ``UNet.py:13``
                     (MaxPool2d(2, 2)).  Pinned by golden vectors captured from
                     the importable reference class (tests/golden/make_golden.py).
* temporal blocks -- ``MultiheadAttention`` / ``TransformerEncoder`` of
                     ``__pycache__/TransformerEncoder.cpython-38.pyc`` (source
                     lines L6-L126 as recorded in the bytecode line table; the
                     ``.py`` is not in the repository, the bytecode is for
                     CPython 3.8 and cannot be imported here).  PARITY UNPINNED
                     BY THE REFERENCE: it ships no test, golden vector or
                     known-answer for these classes.  This restatement follows
                     the recovered specification line by
                     line and is pinned by our own hand-computed KATs and an
                     independent ``scaled_dot_product_attention`` cross-check
                     (tests/test_oracle.py).
* frame token, head, loss -- the composite's own glue (the reference has no 5-D
                     model, classifier or cross-entropy).  Parity
                     unpinned by the reference; defined here.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# conv stage: UNet.py:58-60 + UNet.py:13
# ----------------------------------------------------------------------------
def conv_stage(in_channels, features, name):
    """First Conv+BN+ReLU triple of ``UNet._block`` (UNet.py:54-66, lines 58-60)
    followed by the ``MaxPool2d(kernel_size=2, stride=2)`` of UNet.py:13.  Module
    and state-dict key names follow the reference (``{name}conv1``, ``{name}norm1``)."""
    return nn.Sequential(OrderedDict([
        (name + "conv1", nn.Conv2d(in_channels, features, kernel_size=3, padding=1, bias=False)),
        (name + "norm1", nn.BatchNorm2d(num_features=features)),
        (name + "relu1", nn.ReLU(inplace=True)),
        (name + "pool1", nn.MaxPool2d(kernel_size=2, stride=2)),
    ]))


# ----------------------------------------------------------------------------
# TransformerEncoder.pyc src L6-L89
# ----------------------------------------------------------------------------
class MultiheadAttention(nn.Module):
    def __init__(self, input_dim, num_heads):                      # L7-19
        super().__init__()
        self.input_dim = input_dim
        self.num_heads = num_heads
        self.query_layer = nn.Linear(input_dim, input_dim)           # L12
        self.key_layer = nn.Linear(input_dim, input_dim)             # L13
        self.value_layer = nn.Linear(input_dim, input_dim)           # L14
        self.output_layer = nn.Linear(input_dim, input_dim)          # L15
        self.activation = nn.ReLU()                                  # L17
        self.softmax = nn.Softmax(dim=-1)                            # L18
        self.dropoutLayer = nn.Dropout(0.1)                          # L19

    def __reshape_to_batches__(self, x):                             # L22-37
        batch_size, seq_len, in_feature = x.size()
        sub_dim = in_feature // self.num_heads
        return (x.reshape(batch_size, seq_len, self.num_heads, sub_dim)
                 .permute(0, 2, 1, 3)
                 .reshape(batch_size * self.num_heads, seq_len, sub_dim))

    def __reshape_from_batches__(self, x):                           # L38-45
        batch_size, seq_len, in_feature = x.size()
        batch_size //= self.num_heads
        out_dim = in_feature * self.num_heads
        return (x.reshape(batch_size, self.num_heads, seq_len, in_feature)
                 .permute(0, 2, 1, 3)
                 .reshape(batch_size, seq_len, out_dim))

    def attention(self, q, k, v, mask):                              # L49-62
        dot = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(self.input_dim)   # L51 (Q1: sqrt(d_model))
        if mask is not None:                                         # L54-55
            dot = dot.masked_fill(mask == 0, -1e9)
        weights = self.dropoutLayer(self.softmax(dot))               # L58
        return torch.matmul(weights, v)                              # L61-62

    def forward(self, q, k, v, mask=None):                           # L67-89
        q, k, v = self.query_layer(q), self.key_layer(k), self.value_layer(v)     # L69
        q, k, v = self.activation(q), self.activation(k), self.activation(v)      # L70 (Q2)
        q = self.__reshape_to_batches__(q)                           # L73-75
        k = self.__reshape_to_batches__(k)
        v = self.__reshape_to_batches__(v)
        if mask is not None:                                         # L77-78 (Q4)
            mask = mask.repeat(self.num_heads, 1, 1)
        a = self.attention(q, k, v, mask)                            # L81
        a = self.__reshape_from_batches__(a)                         # L84
        return self.output_layer(a)                                  # L87-89


class TransformerEncoder(nn.Module):
    def __init__(self, input_dim, hidden_dim, num_layers, num_heads, dropout):   # L94-108
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.dropout = dropout
        if input_dim % num_heads != 0:                               # L102-103
            raise ValueError(
                f"Input dimension must be divisible by number of heads. Here, Input dimension = {input_dim}"
                f" is not divisible by number of heads = {num_heads}")
        self.attention_layers = nn.ModuleList(
            [MultiheadAttention(input_dim, num_heads) for _ in range(num_layers)])          # L106
        self.feedforward_layers = nn.ModuleList(
            [nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, input_dim))
             for _ in range(num_layers)])                                                     # L107
        self.layer_norm = nn.ModuleList([nn.LayerNorm(input_dim) for _ in range(num_layers)])  # L108

    def forward(self, input, mask):                                  # L110-126
        for i in range(self.num_layers):                             # L113
            skip1 = input                                            # L114
            input = self.attention_layers[i](input, input, input, mask)   # L115
            input = self.layer_norm[i](input)                        # L116 (Q3: LN before the residual add)
            input = input + skip1                                    # L117
            skip2 = input                                            # L118
            input = self.feedforward_layers[i](input)                # L119
            input = self.layer_norm[i](input)                        # L120 (same LayerNorm instance)
            input = input + skip2                                    # L121
            input = input * math.sqrt(0.5)                           # L122 (Q7)
            input = nn.Dropout(self.dropout)(input)                  # L123 (Q6: fresh module => always active)
        return input                                                 # L126


# ----------------------------------------------------------------------------
# composite (SURVEY.md section 0.3 binding decisions; BASELINE config-2 defaults)
# ----------------------------------------------------------------------------
class TransformerCNNHybridRef(nn.Module):
    """[B,T,3,H,W] (or [B,3,H,W] => T=1) -> logits [B,num_classes]."""

    def __init__(self, in_channels=3, cnn_channels=(32, 64, 128, 256), d_model=512, num_heads=8,
                 num_layers=2, hidden_dim=2048, num_classes=8, dropout=0.0):
        super().__init__()
        chans = (in_channels,) + tuple(cnn_channels)
        for i in range(len(cnn_channels)):
            setattr(self, f"encoder{i + 1}", conv_stage(chans[i], chans[i + 1], f"enc{i + 1}"))
        self.num_stages = len(cnn_channels)
        self.token_proj = nn.Linear(chans[-1], d_model)
        self.encoder = TransformerEncoder(d_model, hidden_dim, num_layers, num_heads, dropout)
        self.head = nn.Linear(d_model, num_classes)

    def forward(self, x, mask=None):
        if x.dim() == 4:
            x = x.unsqueeze(1)
        B, T = x.shape[:2]
        f = x.reshape(B * T, *x.shape[2:])                 # frames folded into the batch axis
        for i in range(self.num_stages):
            f = getattr(self, f"encoder{i + 1}")(f)
        f = f.mean(dim=(2, 3))                             # global average pool -> [B*T, C]
        tok = self.token_proj(f).reshape(B, T, -1)
        enc = self.encoder(tok, mask)
        return self.head(enc.mean(dim=1))


def loss_fn(logits, target):
    return F.cross_entropy(logits, target)


def synthetic_batch(B, T, H, W, num_classes=8, seed=0, device="cpu"):
    """uniform [0,1) fp32 clips + randint labels, seeded."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, T, 3, H, W, generator=g)
    y = torch.randint(0, num_classes, (B,), generator=g)
    return x.to(device), y.to(device)
