from .priors import *  # noqa: F401,F403
from .spatiotemporalpriors import *  # noqa: F401,F403
from .stem_roi import *  # noqa: F401,F403
from .stem_utils import *  # noqa: F401,F403
