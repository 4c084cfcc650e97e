"""Batch-sharded fake-quantization: one process per GPU, the activation split along dim 0.

The quantize/dequantize math is independent per element; only the scale statistic couples the
shards.  So the path shards with NO data-path collective except

  forward   one all-reduce(MAX) of the per-channel (or whole-tensor) abs-max     <= 4 KB
  backward  one all-gather of [scale-gradient partial sums | tie ownership]       <= 8 KB x ranks

over RCCL/xGMI (backend "nccl" on ROCm; "gloo" works for CPU tensors in tests; enable_native_collectives() lets the
C++ node of the sharded quantizer issue both through RCCL's C API on the compute stream).  Both are
latency-bound messages; they sit between the statistic kernel and the quantize kernel (resp. between
the backward kernel and the tiny deposit kernel), so no extra pass over the tensor is made.

Percentile statistics (the default Int8ActPerTensorFloat collects AbsPercentile for its first 300 steps)
shard the same way, as a distributed radix select: every shard histograms one key digit of its own
elements, the 8 KB-per-channel histogram is all-reduced (SUM), every shard picks the same digit; 2 rounds
for 16-bit types, 3 for float32 (sharded_kth_value).  The rank k comes from the GLOBAL element count,
which the first summed histogram already holds -- it is evaluated on the device, nothing goes to the host.

Semantics: the result equals the single-device result on the concatenated batch (a max is exact and
associative, so y is bit-identical); the reference itself has no cross-device reduction
(nn.DataParallel replicas use local statistics).  The statistic's gradient is deposited
where the single-device run would put it: on the first arg-max in batch order, i.e. on the lowest
rank that holds one (per-channel), or evenly over all ties of all shards (whole tensor).  The
scale-gradient sums are combined by the same float64 reduction of the same gathered data on every
rank: bit-identical across ranks and run-to-run.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

_NO_OWNER = float(1 << 30)


def shard_over_batch(quantizer, group=None):
    """Mark an activation quantizer (RescalingIntQuant) as operating on one batch shard of a tensor spread
    over `group` (default: the world group): its fused stats-scaled graph and every statistic module
    inside it (AbsMax, AbsMinMax, NegativeMinOrZero, the percentile family) then reduce over all shards.
    Weight quantizers are replicated, not sharded: do not mark them."""
    if not dist.is_initialized():
        raise RuntimeError('shard_over_batch: torch.distributed is not initialised')
    group = group if group is not None else dist.group.WORLD
    quantizer.bvq_shard_group = group
    for m in quantizer.modules():
        if getattr(m, 'bvq_shardable_stat', False):
            m.bvq_shard_group = group
        elif getattr(m, 'bvq_is_stat', False):
            raise NotImplementedError('%s has no batch-sharded form' % type(m).__name__)
    return quantizer


def enable_native_collectives(group=None) -> bool:
    """Give `group` (default: the world group) a communicator of its own on RCCL's C API, used by the C++ autograd node
    of the batch-sharded quantizer (brevitas_amd/csrc/bvq_autograd.cpp): its all-reduce and all-gather are then ONE
    ncclAllReduce / ncclAllGather call each on the compute stream -- no c10d work object (~17-20 us of host time per
    call), no hop to RCCL's side stream and back (two event waits around a 4 KB message).  Collective: every rank of
    the group calls it, with its device current.  The communicator is checked against torch.distributed before it is
    kept: an all-reduce(MAX) and an all-gather of rank-dependent data must equal c10d's results on every rank.
    -> True (in use) / False (not available or failed the check: the node keeps issuing its collectives through c10d)"""
    from brevitas_amd.core.quant import _fused
    if not dist.is_initialized():
        raise RuntimeError('enable_native_collectives: torch.distributed is not initialised')
    group = group if group is not None else dist.group.WORLD
    fast = _fused._fast_module()
    if not fast or dist.get_backend(group) != 'nccl' or not torch.cuda.is_available():
        return False
    rank, world = world_of(group)
    name = group.group_name
    dev = torch.device('cuda', torch.cuda.current_device())
    def agree(flag: bool) -> bool:  # True only if every rank says so (a torch.distributed collective: same order on every rank)
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(t.item() > 0.5)

    box = [None]
    if rank == 0:
        try:
            box[0] = fast.rccl_unique_id()
        except Exception:
            box[0] = None
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0), group=group)
    if box[0] is None:
        return False  # (the same answer on every rank: rank 0's)
    ok = True
    try:
        fast.rccl_comm_init(name, box[0], world, rank)
    except Exception:  # an RCCL error at set-up: keep c10d
        ok = False
    if not agree(ok):
        try:
            fast.rccl_comm_drop(name)
        except Exception:
            pass
        return False
    # the node's two calls on rank-dependent data against torch.distributed's: the reference results first (every rank,
    # outside any try), then the direct calls
    st = torch.cuda.current_stream(dev).cuda_stream
    a = (torch.arange(64, device=dev, dtype=torch.float32) * (1 + rank) % 7) - rank
    want = a.clone()
    dist.all_reduce(want, op=dist.ReduceOp.MAX, group=group)
    m = torch.arange(16, device=dev, dtype=torch.float64) + 1000.0 * rank
    wantg = torch.empty(16 * world, device=dev, dtype=torch.float64)
    dist.all_gather_into_tensor(wantg, m, group=group)
    try:
        got = torch.empty_like(wantg)
        fast.rccl_all_reduce_max(name, a, st)
        fast.rccl_all_gather_f64(name, m, got, st)
        ok = bool(torch.equal(a, want)) and bool(torch.equal(got, wantg))
    except Exception:
        ok = False
    ok = agree(ok)
    if not ok:
        try:
            fast.rccl_comm_drop(name)
        except Exception:
            pass
    return ok


def disable_native_collectives(group=None) -> None:
    """drop the native communicator of `group` (before destroy_process_group, or to go back to c10d)"""
    from brevitas_amd.core.quant import _fused
    group = group if group is not None else dist.group.WORLD
    fast = _fused._fast_module()
    if fast:
        fast.rccl_comm_drop(group.group_name)


def world_of(group) -> Tuple[int, int]:
    return dist.get_rank(group), dist.get_world_size(group)


def sync_stat_max(stat_f32: Tensor, group) -> Tensor:
    """in-place all-reduce(MAX) of a float32 statistic; returns it"""
    if dist.get_world_size(group) > 1:
        dist.all_reduce(stat_f32, op=dist.ReduceOp.MAX, group=group)
    return stat_f32


def sync_stat_min(stat_f32: Tensor, group) -> Tensor:
    if dist.get_world_size(group) > 1:
        dist.all_reduce(stat_f32, op=dist.ReduceOp.MIN, group=group)
    return stat_f32


KTH_MAX_COUNT = 2 ** 32 - 1


def sharded_kth_value(steps, group) -> Tensor:
    """k-th value of the concatenation of all shards.  `steps`: brevitas_amd._native.KthSelectSteps over
    this shard (or any object with its begin / hist(p) / pick(p) / finish and `passes`)."""
    # the digit counters are 32-bit and the all-reduced ones count every shard's elements (include/bvq.h):
    # refuse a channel that could overflow them instead of selecting a wrong value silently
    world = dist.get_world_size(group)
    local = getattr(steps, 'per_channel', 0)
    if local * world > KTH_MAX_COUNT:
        raise ValueError('sharded_kth_value: %d elements per channel on this shard x %d shards exceeds the '
                         '32-bit digit counters of the radix select (limit 2^32 - 1 over all shards)' % (local, world))
    steps.begin()
    for p in range(steps.passes):
        h = steps.hist(p)
        if dist.get_world_size(group) > 1:
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        steps.pick(p)
    return steps.finish()


def sync_backward(ds_local: Tensor, tie_info: Tensor, channels: int, group, first_only: bool = False
                  ) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """Combine the shards' backward bookkeeping with ONE all-gather.

    ds_local : float32 [channels]   this shard's partial sums of the scale gradient
    tie_info : int64 buffer written by the backward kernel (include/bvq.h, bvq_stat_tie_scan):
               channels > 1: word c = first local position attaining the statistic, or -1
               channels == 1: word 0 = number of local ties -- or, with first_only (the statistic's
               gradient goes to ONE element: kthvalue), the first local position like channels > 1
    returns (ds_total float32 [channels], tie_info with non-owned channels disabled,
             total_ties int64 [1] or None)
    """
    rank, world = world_of(group)
    per_channel = channels > 1 or first_only
    if ds_local.is_cuda:
        # the same protocol with the packing / unpacking in one launch each (include/bvq.h, bvq_shard_pack)
        from brevitas_amd import _native as nat
        mine = nat.shard_pack(ds_local.contiguous(), tie_info, channels, rank, per_channel)
        if world > 1:
            flat = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(flat, mine, group=group)
        else:
            flat = mine
        ds_total, total = nat.shard_unpack(flat, world, channels, rank, per_channel, tie_info)
        return ds_total, tie_info, total
    if per_channel:
        has = tie_info[:channels] >= 0
        key = torch.where(has, torch.full_like(ds_local, float(rank), dtype=torch.float64),
                          torch.full_like(ds_local, _NO_OWNER, dtype=torch.float64))
    else:
        key = tie_info[:1].to(torch.float64)
    mine = torch.stack([ds_local.to(torch.float64), key])  # [2, channels]
    if world > 1:
        flat = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(flat, mine.reshape(-1).contiguous(), group=group)
        allr = flat.reshape((world,) + tuple(mine.shape))
    else:
        allr = mine.unsqueeze(0)
    # one float64 reduction over the rank axis of identical data on every rank: same bits everywhere
    ds_total = allr[:, 0].sum(dim=0).to(torch.float32)
    if per_channel:
        owner = allr[:, 1].min(dim=0).values
        out = tie_info.clone()
        out[:channels] = torch.where(owner == float(rank), tie_info[:channels],
                                     torch.full_like(tie_info[:channels], -1))
        return ds_total, out, None
    total = allr[:, 1].sum(dim=0).to(torch.int64).reshape(1)
    return ds_total, tie_info, total
