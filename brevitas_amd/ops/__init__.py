from . import autograd_ste_ops  # noqa: F401
