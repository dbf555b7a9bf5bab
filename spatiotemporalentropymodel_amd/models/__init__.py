from .priors import *  # noqa: F401,F403
from .spatiotemporalpriors import *  # noqa: F401,F403
