"""CPU pin of the oracle on the reference's resolved module graphs (tests/golden/{quant_graphs,shifted,
learned_bw}.npz, produced by the reference's own RescalingIntQuant graphs): given the scale / zero-point /
bit width the reference computed, the oracle's quantize-dequantize reproduces the reference's y bit for bit,
and its dx wherever no statistic's gradient is deposited.  (The device path is checked against the same
files by the -m gpu tests; this keeps the checker itself honest on every graph family.)"""
import numpy as np
import pytest

import golden_util as G


@pytest.fixture(scope='module')
def orc():
    import oracle
    oracle.build()
    return oracle


def _cases():
    out = []
    for c in G.load('quant_graphs'):
        g = c['graph']
        if g == 'weight_per_channel':
            bw = c['bit_width']
            out.append((c, -(2 ** (bw - 1)) + 1, 2 ** (bw - 1) - 1, True, 'wpc-%s-%s' % (c['tag'], c['dtype'])))
        elif g in ('act_runtime_stats', 'act_param_from_stats', 'act_parameter_scale'):
            out.append((c, -128, 127, False, '%s-%s-%s-%s' % (g, c.get('tag'), c['dtype'], c.get('step'))))
    for c in G.load('shifted'):
        if c['graph'] == 'shifted_weight':
            out.append((c, 0, 255, True, 'shw-%s-%s' % (c['tag'], c['dtype'])))
        elif c['graph'] == 'shifted_act':
            out.append((c, 0, 255, False, 'sha-%s-%s' % (c['dtype'], c['step'])))
    for c in G.load('learned_bw'):
        if c['graph'] == 'weight':
            out.append((c, -7, 7, True, 'lbw-w-%s' % c['dtype']))
        elif c['graph'] == 'act':
            b = c['bits']
            out.append((c, -(2 ** (b - 1)), 2 ** (b - 1) - 1, False, 'lbw-a%d-%s' % (b, c['dtype'])))
    return out


CASES = _cases()


@pytest.mark.parametrize('c,qmin,qmax,clamp_ste,name', CASES, ids=[t[4] for t in CASES])
def test_oracle_reproduces_graph_outputs(orc, c, qmin, qmax, clamp_ste, name):
    x = c.arr('x')
    scale = c.arr('scale')
    zp = c.arr('zp') if c.has('zp') else np.zeros(1, np.float32)
    zp_dt = c.dt('zp') if c.has('zp') else orc.F32
    if c['dtypes']['y'] != c['dtypes']['x']:
        pytest.skip('the reference promoted the output (float32 scale tensor against a 16-bit input)')
    pc = scale.size > 1 or zp.size > 1
    if pc:
        ch = max(scale.size, zp.size)
        cd = [i for i, s in enumerate(x.shape) if s == ch and (scale.size == 1 or scale.reshape(-1).size == ch)]
        sshape = scale.shape if scale.size > 1 else zp.shape
        sshape = (1,) * (x.ndim - len(sshape)) + tuple(sshape)
        cd = [i for i, s in enumerate(sshape) if s != 1][0]
        outer, inner = int(np.prod(x.shape[:cd])), int(np.prod(x.shape[cd + 1:]))
    else:
        ch, outer, inner = 1, 1, x.size
    d = orc.make_desc(outer, ch, inner, c.dt('x'), c.dt('y'), c.dt('scale'), zp_dt, scale_per_channel=scale.size > 1,
                      zp_per_channel=zp.size > 1, qmin=float(qmin), qmax=float(qmax), clamp_ste=clamp_ste)
    y, codes = orc.fakequant_fwd(d, x.reshape(-1), scale.reshape(-1), zp.reshape(-1))
    assert np.array_equal(y, c.arr('y').reshape(-1))
    assert codes.min() >= qmin and codes.max() <= qmax
    dx, _, _ = orc.fakequant_bwd(d, c.arr('g').reshape(-1), x.reshape(-1), scale.reshape(-1), zp.reshape(-1))
    want = c.arr('dx').reshape(-1)
    bad = np.nonzero(dx != want)[0]
    # elements holding a statistic (arg-max / min / max / percentile of a channel) also receive its gradient
    assert bad.size <= 2 * ch + 4, (bad.size, ch)
