"""CPU oracle of the fake-quantization hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See bvq_oracle.c for the restatement and its parity pin.
"""
from .bvq_oracle import *  # noqa: F401,F403
