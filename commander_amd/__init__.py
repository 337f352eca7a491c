"""commander_amd -- MI355X-native Gibbs amplitude-sampling path for Commander3.

The product is ``libcmdr_hip.so`` (hand-written gfx950 HIP kernels behind the C ABI of ``include/cmdr_hip.h``);
this package is the thin Python host mirror used by the tests and ``bench.py``.  There is no CPU fallback: every
compute call raises ``CmdrError`` when the HIP library or a GPU is missing.
"""
from .lib import CmdrError, build_library, device_count  # noqa: F401
from .lib import lib as get_lib  # noqa: F401
from .sht import ShtPlan, JOB_Y, JOB_Yt, JOB_YtW, JOB_WY  # noqa: F401
from .cr import CRContext, build_context  # noqa: F401
from . import lib as _lib_module  # noqa: F401,E402
