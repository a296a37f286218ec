from xgnn_amd.common import *  # noqa: F401,F403
from xgnn_amd.common import _basics  # noqa: F401
