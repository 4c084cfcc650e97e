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


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16], ids=['bf16', 'f32', 'f16'])
@pytest.mark.parametrize('channels', [1, 5])
def test_stepwise_select_equals_kth_value(nccl_world1, dtype, channels):
    """bvq_kth_begin/hist/pick/finish with the histogram routed through an RCCL all-reduce and the rank
    derived on the device == bvq_kth_value with the host-computed rank == torch.kthvalue"""
    import math

    from brevitas_amd import _native as nat
    from brevitas_amd.distributed import sharded_kth_value
    torch.manual_seed(123456)
    outer, inner = 7, 333
    x = torch.randn(outer, channels, inner, device=DEV).to(dtype)
    x[0, 0, :40] = 0.5
    n = outer * inner
    rows = x.permute(1, 0, 2).reshape(channels, -1).float()
    for abs_key in (True, False):
        src = rows.abs() if abs_key else rows
        for rule, qs in ((nat.KTH_HIGH, (99.999, 50.0, 0.3)), (nat.KTH_LOW, (0.001, 25.0, 100.0))):
            for q in qs:
                k = int(math.floor(.01 * q * n + 0.5)) if rule == nat.KTH_HIGH else int(math.ceil(.01 * q * n))
                steps = nat.KthSelectSteps(x.reshape(-1), outer, channels, inner, abs_key, rule, q)
                got = sharded_kth_value(steps, nccl_world1)
                want = nat.kth_value(x.reshape(-1), k, outer, channels, inner, abs_key)
                assert torch.equal(got, want), (abs_key, rule, q)
                assert torch.equal(got.float(), src.kthvalue(k, dim=1).values)
    # explicit rank through the stepwise entry points
    steps = nat.KthSelectSteps(x.reshape(-1), outer, channels, inner, True, nat.KTH_EXPLICIT, 0.0, k=17)
    assert torch.equal(sharded_kth_value(steps, nccl_world1),
                       nat.kth_value(x.reshape(-1), 17, outer, channels, inner, True))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'f32'])
@pytest.mark.parametrize('name', ['Int8ActPerTensorFloat', 'ShiftedUint8ActPerTensorFloat',
                                  'Int8ActPerTensorFloat-max', 'Int8ActPerTensorFixedPoint'])
def test_sharded_named_act_quantizers_equal_unsharded(nccl_world1, name, dtype):
    """percentile / min-max / abs-max statistics of the named activation quantizers through the sharded
    statistic modules (collectives on RCCL) == the unsharded modules, over the collection steps, the switch
    to the learned scale and one learned step"""
    import brevitas_amd.quant as Q
    from brevitas_amd.distributed import shard_over_batch

    def build():
        if name == 'Int8ActPerTensorFloat':
            return Q.Int8ActPerTensorFloat(collect_stats_steps=2)
        if name == 'Int8ActPerTensorFloat-max':
            return Q.Int8ActPerTensorFloat(collect_stats_steps=2, scaling_stats_op='max')
        if name == 'Int8ActPerTensorFixedPoint':
            return Q.Int8ActPerTensorFixedPoint(collect_stats_steps=2)
        return Q.ShiftedUint8ActPerTensorFloat(collect_stats_steps=2)

    view = torch.int16 if dtype == torch.bfloat16 else torch.int32
    qa, qb = build().to(DEV), shard_over_batch(build().to(DEV), nccl_world1)
    qa.train(), qb.train()
    torch.manual_seed(123456)
    for step in range(4):
        x = (torch.randn(4, 6, 9, 9, device=DEV) * (1.0 + 0.3 * step) + 0.2).to(dtype)
        g = torch.randn_like(x)
        outs = []
        for q in (qa, qb):
            xi = x.clone().requires_grad_(True)
            q.zero_grad()
            y, scale, zp, bw = q(xi)
            y.backward(g)
            outs.append((y.detach(), scale.detach().float(), zp.detach().float(), xi.grad))
        (y0, s0, z0, dx0), (y1, s1, z1, dx1) = outs
        assert torch.equal(y0.view(view), y1.view(view)) and torch.equal(s0, s1) and torch.equal(z0, z1), step
        diff = (dx0.view(view) != dx1.view(view)).reshape(-1).nonzero().reshape(-1)
        assert diff.numel() <= 4, (step, diff.numel())  # the elements holding a statistic: float32 route of gsum
        if diff.numel():
            a, b = dx0.reshape(-1)[diff].float(), dx1.reshape(-1)[diff].float()
            assert bool(((a - b).abs() <= 2.0 ** -6 * (a.abs() + b.abs() + 1e-3)).all())


@pytest.mark.parametrize('per_channel', [True, False], ids=['per_channel', 'per_tensor'])
def test_shard_pack_unpack_kernels_for_three_ranks(per_channel):
    """bvq_shard_pack / bvq_shard_unpack with the messages of three simulated ranks (no collective needed) against
    the torch-op statement of the same protocol (the one the gloo test runs on CPU tensors)"""
    from brevitas_amd import _native as nat
    torch.manual_seed(123456)
    world, C = 3, 37 if per_channel else 1
    ds = [torch.randn(C, device=DEV) * 10 for _ in range(world)]
    if per_channel:
        infos = [torch.randint(-1, 50, (C,), device=DEV, dtype=torch.int64).clamp_min(-1) for _ in range(world)]
        infos[0][:5] = -1   # channels nobody on rank 0 holds
        infos[1][:3] = -1
        infos[2][:2] = -1   # channels 0,1: nobody at all
    else:
        infos = [torch.tensor([k, 0, 11, 12], device=DEV, dtype=torch.int64) for k in (2, 0, 5)]
    msgs = [nat.shard_pack(ds[r], infos[r], C, r, per_channel) for r in range(world)]
    gathered = torch.cat(msgs)
    all_ = gathered.view(world, 2, C)
    want_ds = all_[:, 0].sum(dim=0).to(torch.float32)
    for r in range(world):
        info = infos[r].clone()
        ds_total, total = nat.shard_unpack(gathered, world, C, r, per_channel, info)
        assert torch.allclose(ds_total, want_ds, rtol=1e-6, atol=1e-6)
        if per_channel:
            owner = all_[:, 1].min(dim=0).values
            want = torch.where(owner == float(r), infos[r][:C], torch.full_like(infos[r][:C], -1))
            assert torch.equal(info[:C], want) and total is None
        else:
            assert int(total) == 7 and torch.equal(info, infos[r])
    # every rank computes the same bits
    outs = [nat.shard_unpack(gathered, world, C, r, per_channel, infos[r].clone())[0] for r in range(world)]
    assert all(torch.equal(outs[0].view(torch.int32), o.view(torch.int32)) for o in outs)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16], ids=['bf16', 'f32', 'f16'])
def test_scale_from_stat_equals_the_three_torch_ops(dtype):
    from brevitas_amd import _native as nat
    torch.manual_seed(1)
    stat32 = (torch.rand(64, device=DEV) * 5).to(dtype).float()  # values of dtype, as a max of such values is
    stat32[3] = 0.0
    stat32[4] = float('nan')
    stat, scale = nat.scale_from_stat(stat32, dtype, 1e-10, 128.0, dtype)
    want_stat = stat32.to(dtype)
    want_scale = torch.clamp_min(want_stat, 1e-10) / torch.tensor(128.0, device=DEV)
    view = torch.int16 if dtype != torch.float32 else torch.int32
    assert torch.equal(stat.view(view), want_stat.view(view))
    assert torch.equal(scale.view(view), want_scale.to(dtype).view(view))
    # 0-dim promotion: float32 scale of a 16-bit statistic
    stat, scale = nat.scale_from_stat(stat32[:1].contiguous(), dtype, None, 127.0, torch.float32)
    assert scale.dtype == torch.float32 and float(scale) == float(stat32[0].to(dtype).float() / 127.0)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32], ids=['bf16', 'f16', 'f32'])
def test_wide_stepwise_select_equals_kth_value(nccl_world1, dtype):
    """bvq_kthw_* (15-bit first digit, whole-tensor statistic) through the RCCL all-reduce == bvq_kth_value ==
    torch.kthvalue: both key kinds, both rank rules, an explicit rank, a buffer that does not start on a 16-byte
    boundary, a tiny one"""
    import math

    from brevitas_amd import _native as nat
    from brevitas_amd.distributed import sharded_kth_value
    torch.manual_seed(123456)
    base = torch.randn(70001, device=DEV).to(dtype)
    base[100:180] = 0.5
    for x in (base, base[3:], base[1:40], base[5:6]):
        n = x.numel()
        for abs_key in (True, False):
            src = x.float().abs() if abs_key else x.float()
            assert nat.KthWideSteps(x, abs_key, nat.KTH_HIGH, 50.0).passes == (1 if abs_key and dtype != torch.float32 else 2)
            for rule, qs in ((nat.KTH_HIGH, (99.999, 50.0, 0.3)), (nat.KTH_LOW, (0.001, 25.0, 100.0))):
                for q in qs:
                    k = int(math.floor(.01 * q * n + 0.5)) if rule == nat.KTH_HIGH else int(math.ceil(.01 * q * n))
                    k = min(max(k, 1), n)
                    got = sharded_kth_value(nat.KthWideSteps(x, abs_key, rule, q), nccl_world1)
                    assert torch.equal(got.float().reshape(()), src.kthvalue(k).values), (n, abs_key, rule, q)
            k = min(17, n)
            got = sharded_kth_value(nat.KthWideSteps(x, abs_key, nat.KTH_EXPLICIT, 0.0, k=k), nccl_world1)
            assert torch.equal(got.float().reshape(()), src.kthvalue(k).values)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'f32'])
@pytest.mark.parametrize('split', [0, 1, 4099, 50000], ids=lambda s: 'split%d' % s)
def test_wide_stepwise_select_over_two_shards_on_one_device(dtype, split):
    """the protocol with two shards held by one process: the counters of both shards are summed by hand where the
    all-reduce would do it -> the k-th value of the concatenation, from either shard's workspace"""
    from brevitas_amd import _native as nat
    torch.manual_seed(123456 + split)
    full = torch.randn(50000, device=DEV).to(dtype)
    full[[3, 49000]] = 2.5
    full[200:260] = -0.25
    shards = [full[:split].contiguous(), full[split:].contiguous()]
    for abs_key, rule, q in ((True, nat.KTH_HIGH, 99.999), (True, nat.KTH_HIGH, 50.0), (False, nat.KTH_LOW, 0.1),
                             (False, nat.KTH_HIGH, 75.0)):
        steps = [nat.KthWideSteps(s, abs_key, rule, q) for s in shards]
        for s in steps:
            s.begin()
        for p in range(steps[0].passes):
            hs = [s.hist(p) for s in steps]
            total = hs[0] + hs[1]
            for h in hs:
                h.copy_(total)
            for s in steps:
                s.pick(p)
        vals = [s.finish() for s in steps]
        n = full.numel()
        import math
        k = int(math.floor(.01 * q * n + 0.5)) if rule == nat.KTH_HIGH else int(math.ceil(.01 * q * n))
        src = full.float().abs() if abs_key else full.float()
        want = src.kthvalue(min(max(k, 1), n)).values
        assert torch.equal(vals[0].float().reshape(()), want) and torch.equal(vals[1], vals[0]), (abs_key, rule, q)


def test_c10d_collectives_are_never_captured(nccl_world1):
    """bench.graphed_run refuses a sharded step whose collectives go through torch.distributed: its watchdog thread may
    query an event recorded in the capturing stream (hipErrorCapturedEvent -> terminate(): seen on this image).  The
    captured form of the sharded step is test_native_collectives_equal_c10d's, with direct RCCL calls."""
    import bench
    job = bench.Job('act_pc', torch.bfloat16, torch.device(DEV), nccl_world1, 0, act_shape=(4, 64, 28, 28))
    elapsed, note = bench.graphed_run(job, steps=2, warmup=1, world=1, device=torch.device(DEV), group=nccl_world1)
    assert elapsed is None and 'not captured' in note


def test_native_collectives_equal_c10d(nccl_world1):
    """the sharded quantizer with its two collectives issued through RCCL's C API from the C++ node
    (brevitas_amd.distributed.enable_native_collectives: communicator set-up, the check against c10d, the node's calls)
    equals the same quantizer on c10d and the unsharded one bit for bit; it also replays from a HIP graph"""
    import bench
    from brevitas_amd.core.quant import _fused
    from brevitas_amd.distributed import disable_native_collectives, enable_native_collectives
    if not _fused._fast_module():
        pytest.skip('the C++ autograd node is not built')
    torch.manual_seed(7)
    x = torch.randn(8, 32, 28, 28, device=DEV, dtype=torch.bfloat16)
    g = torch.randn_like(x)

    def run(group, per_channel=True):
        q = bench.build_quantizer(32, per_channel, torch.device(DEV), group)
        xi = x.clone().requires_grad_(True)
        y, scale = q(xi)[:2]
        y.backward(g)
        return y.detach(), scale.detach(), xi.grad

    plain, plain_t = run(None), run(None, False)
    c10d = run(nccl_world1)
    assert enable_native_collectives(nccl_world1) is True
    try:
        assert _fused._fast_module().rccl_comm_active(nccl_world1.group_name)
        native = run(nccl_world1)
        for a, b, c, what in zip(plain, c10d, native, ('y', 'scale', 'dx')):
            assert torch.equal(a, b) and torch.equal(a, c), what
        # a whole-tensor statistic: the forward's all-reduce is the direct call, the backward's bookkeeping stays on
        # torch.distributed (the Python route) -- the two communicators side by side
        for a, b, what in zip(plain_t, run(nccl_world1, False), ('y', 'scale', 'dx')):
            assert torch.equal(a, b), ('per-tensor', what)
        job = bench.Job('act_pc', torch.bfloat16, torch.device(DEV), nccl_world1, 0, act_shape=(4, 64, 28, 28))
        elapsed, note = bench.graphed_run(job, steps=5, warmup=2, world=1, device=torch.device(DEV), group=nccl_world1)
        assert elapsed is not None and 'verified' in note, note
    finally:
        disable_native_collectives(nccl_world1)
    assert not _fused._fast_module().rccl_comm_active(nccl_world1.group_name)
