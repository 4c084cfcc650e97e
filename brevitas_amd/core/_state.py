"""Checkpoint tolerance shared by the stateful quantizer modules.

Brevitas modules accept checkpoints that lack some of their entries: buffers that are never saved, the
`training` attribute older TorchScript checkpoints did not carry, and -- when the user sets
BREVITAS_IGNORE_MISSING_KEYS=1 to start from a float checkpoint -- every learned quantization parameter
(B/core/utils.py:41-63, B/core/scaling/standalone.py:137-152, B/core/stats/stats_wrapper.py:69-80, ...).
Instead of one hand-written load hook per class, a module lists its entries here."""
import brevitas_amd.config as config


class TolerantLoad:
    """Mixin (put it before torch.nn.Module in the bases)."""

    #: own entries that never appear in a state dict
    bvq_never_saved = ('training',)
    #: own entries a float checkpoint lacks; forgiven when config.IGNORE_MISSING_KEYS is set
    bvq_float_checkpoint_ok = ()

    def _load_from_state_dict(self, state_dict, prefix, *hook_args):
        super()._load_from_state_dict(state_dict, prefix, *hook_args)
        self.bvq_forgive_missing(prefix, hook_args[2])

    def bvq_forgive_missing(self, prefix, missing_keys):
        forgiven = list(self.bvq_never_saved)
        if config.IGNORE_MISSING_KEYS:
            forgiven += list(self.bvq_float_checkpoint_ok)
        for name in forgiven:
            key = prefix + name
            if key in missing_keys:
                missing_keys.remove(key)


class CollectThenLearn(TolerantLoad):
    """The protocol shared by ParameterFromRuntimeStatsScaling and ParameterFromRuntimeZeroPoint
    (B/core/scaling/standalone.py:155-298, B/core/zero_point.py:86-183): for the first `collect_stats_steps`
    training forwards a statistic of the input is averaged into `buffer`; the step after, the average becomes
    the initial value of the learned parameter `value`, which is used from then on.

    Checkpoints: `buffer` is never saved; before anything was collected neither is `value`; during collection
    the running average is saved AS `value`; loading a checkpoint that has `value` ends the collection.
    Subclasses provide `bvq_collected()` -- the running average in the form `value` is kept in."""

    bvq_never_saved = ('training', 'buffer')
    bvq_float_checkpoint_ok = ('value',)

    def bvq_init_collection(self, steps: int, shape, fill: float, momentum):
        import torch
        assert steps > 0, 'Steps should be more than 0'
        self.collect_stats_steps = steps
        self.counter = 0
        self.momentum = momentum
        self.register_buffer('buffer', torch.full(shape, fill))
        self.value = torch.nn.Parameter(torch.full(shape, fill))

    def bvq_fold(self, stat, first_op) -> None:
        """one more batch statistic into the running average; `first_op(buffer, stat)` seeds it"""
        from brevitas_amd.core.utils import inplace_momentum_update
        seen = self.counter
        if seen == 0:
            first_op(self.buffer, stat)
        else:
            inplace_momentum_update(self.buffer, stat, self.momentum, seen, seen + 1)
        self.counter = seen + 1

    def bvq_collected(self):
        return self.buffer

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        entries = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        entries.pop(prefix + 'buffer')
        if self.counter == 0:
            entries.pop(prefix + 'value')
        elif self.counter <= self.collect_stats_steps:
            entries[prefix + 'value'] = self.bvq_collected()
        return entries

    def _load_from_state_dict(self, state_dict, prefix, *hook_args):
        if prefix + 'value' in state_dict:  # saved during or after collection: carry on as learned
            self.counter = self.collect_stats_steps + 1
        super()._load_from_state_dict(state_dict, prefix, *hook_args)
