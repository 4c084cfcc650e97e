from .ops import *  # noqa: F401,F403
from .ops_ste import *  # noqa: F401,F403
