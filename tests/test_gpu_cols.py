"""Column-mapped kernels (channel axis last or nearly last: NHWC, [tokens, hidden], 7x7 maps) against the
row-mapped path they replace on those layouts (BVQ_COLS=0 selects it; it is the path pinned to the
reference's golden vectors) and against plain torch reductions: bit-identical statistics, outputs and
gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}

SHAPES = [  # (outer, channels, inner)
    (300, 512, 1),    # [tokens, hidden] / NHWC: one full strip per row
    (257, 64, 1),     # short rows: several rows per wave pass
    (130, 24, 1),     # 3 chunks per row (bf16): 21 rows per pass, idle lanes
    (65, 1000, 1),    # two strips, the second one partly empty
    (33, 16, 49),     # 7x7 maps
    (40, 6, 12),      # small inner, odd sizes
    (5000, 8, 2),     # many rows: several row blocks
]


def bits(t):
    return t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_cols_absmax(dn, shape):
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    torch.manual_seed(123456)
    x = (torch.randn(outer, ch, inner, device=DEV) * 3).to(DT[dn])
    x[0, 1, 0] = float('nan') if dn == 'f32' else 7.0
    for pre in (0, 1):
        stat = nat.stats(nat.STAT_ABSMAX, x.reshape(-1), outer, ch, inner, pre_op=pre)
        src = torch.relu(x) if pre else x
        want = src.abs().amax(dim=(0, 2))
        assert torch.equal(bits(stat), bits(want)), pre
        st2, scale = nat.absmax_scale(x.reshape(-1), outer, ch, inner, 1e-10, 128.0, DT[dn], pre)
        assert torch.equal(bits(st2), bits(want))
