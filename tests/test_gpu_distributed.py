"""The batch-sharded path on the real device with RCCL (backend "nccl"), world size 1: every
collective of brevitas_amd.distributed runs on GPU tensors through RCCL, and the result must equal
the unsharded quantizer bit for bit.  (World sizes > 1 need several GPUs: the protocol itself is
covered by the world-size-2 gloo test, the kernels by the other -m gpu tests.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def nccl_world1():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    yield dist.group.WORLD
    dist.destroy_process_group()


@pytest.mark.parametrize('per_channel', [True, False], ids=['per_channel', 'per_tensor'])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'f32'])
def test_sharded_path_equals_unsharded(nccl_world1, per_channel, dtype):
    from bench import build_quantizer
    torch.manual_seed(123456)
    x = torch.randn(8, 16, 14, 14, device=DEV, dtype=dtype)
    x[1, 3, 2, 2] = 7.0
    x[5, 3, 1, 1] = -7.0  # a +-max tie: the first one gets the deposit
    g = torch.randn_like(x)
    outs = []
    for group in (None, nccl_world1):
        q = build_quantizer(16, per_channel, torch.device(DEV), group)
        xi = x.clone().requires_grad_(True)
        y, scale, zp, bw = q(xi)
        # use the scale output too, so that an external scale gradient enters the exchange
        (y * 1.0).backward(g)
        outs.append((y.detach(), scale.detach(), xi.grad, q.scaling_impl.runtime_stats.running_stats.clone()))
    (y0, s0, dx0, r0), (y1, s1, dx1, r1) = outs
    view = torch.int16 if dtype == torch.bfloat16 else torch.int32
    assert torch.equal(y0.view(view), y1.view(view)) and torch.equal(s0, s1) and torch.equal(r0, r1)
    # dx identical except (at most) the deposit positions, whose reduced sum takes a different route
    diff = (dx0.view(view) != dx1.view(view)).reshape(-1).nonzero().reshape(-1)
    assert diff.numel() <= (16 if per_channel else 2)
    if diff.numel():
        a, b = dx0.reshape(-1)[diff].float(), dx1.reshape(-1)[diff].float()
        assert bool(((a - b).abs() <= 2.0 ** -6 * (a.abs() + b.abs() + 1e-3)).all())
