"""QKVAttention (reference models/networks/modules.py:538-547) at the token counts of the dense net's attention levels:
the split-keys launch (round 5: the four waves of a block share 32 queries and split the keys; csrc/ofx_dense.hip) against
the formula in float64 and against the one-wave-per-32-queries launch it replaces for T >= 256."""
import math

import pytest
import torch

from test_gpu_fullwidth import dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _ref(qkv, B, T, heads):
    """modules.py:540-547 in float64: qkv [B * T, 3 C] rows, channel = head * 3 ch + {q | k | v}."""
    C = qkv.shape[1] // 3
    ch = C // heads
    x = qkv.double().view(B, T, heads, 3, ch)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]                      # [B, T, heads, ch]
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum('bthc,bshc->bhts', q * scale, k * scale)
    w = torch.softmax(w, dim=-1)
    a = torch.einsum('bhts,bshc->bthc', w, v)
    return a.reshape(B * T, C)


@pytest.mark.parametrize('B,T,heads,ch', [(8, 512, 4, 32), (1, 512, 4, 16), (3, 288, 2, 24), (2, 256, 4, 32), (2, 256, 2, 64),
                                          (4, 64, 4, 64)])
def test_attention_split_keys(B, T, heads, ch):
    from octfusion_amd import _lib, ops
    g = torch.Generator().manual_seed(T * 7 + ch)
    C = heads * ch
    qkv = torch.randn(B * T, 3 * C, generator=g) * 1.5
    ref = _ref(qkv, B, T, heads)
    x = qkv.to(dev())
    try:
        y = ops.attention(x, B, T, heads)
        e = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
        assert e < 2e-6, e
        _lib.call('ofx_set_attention_split', 0)
        y0 = ops.attention(x, B, T, heads)
        e0 = float((y0.double().cpu() - ref).abs().max() / ref.abs().max())
        assert e0 < 2e-6, e0
        # same arithmetic, the sums over the keys grouped by wave: last-bit differences only
        assert float((y - y0).abs().max()) <= 2e-6 * float(y0.abs().max())
        # deterministic
        _lib.call('ofx_set_attention_split', 1)
        assert torch.equal(ops.attention(x, B, T, heads), y)
    finally:
        _lib.call('ofx_set_attention_split', 1)
