from .calibrate import DisableEnableQuantization, calibration_mode, finalize_collect_stats

__all__ = ['calibration_mode', 'finalize_collect_stats', 'DisableEnableQuantization']
