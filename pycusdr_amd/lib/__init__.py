from .filters import rrcosfilter, gaussianFilter  # noqa: F401
from .gmskmod import gmskMod  # noqa: F401
