"""spatiotemporalentropymodel_amd: the STEM hot path of mmSir/SpatioTemporalEntropyModel on MI355X.

Importing the package does not load the HIP library (so CPU-only tooling can import weights/fixtures);
the first device op does, and raises if it is missing: there is no CPU fallback.
"""
__version__ = "0.1.0"

from .entropy_models import available_entropy_coders, get_entropy_coder, set_entropy_coder  # noqa: F401


def install_compressai_alias():
    """Expose this package under the reference's import names so that its scripts run unchanged:
    `from compressai.zoo import models`, `from compressai.models.spatiotemporalpriors import *`,
    `from compressai.ans import BufferedRansEncoder, RansDecoder`, `import compressai; compressai.set_entropy_coder(..)`
    (stem/trainSTEM.py:3-4, stem/evalSTEM.py:13-16,268-270)."""
    import sys
    import types

    from . import entropy_models, layers, ops, zoo
    import importlib

    from .models import priors, spatiotemporalpriors, stem_utils, utils
    stem_roi = importlib.import_module(".models.stem_roi", __name__)      # the package attribute of that name is the class

    root = types.ModuleType("compressai")
    root.__path__ = []
    root.set_entropy_coder, root.get_entropy_coder = set_entropy_coder, get_entropy_coder
    root.available_entropy_coders = available_entropy_coders
    ans = types.ModuleType("compressai.ans")
    ans.RansEncoder, ans.RansDecoder = entropy_models.RansEncoder, entropy_models.RansDecoder
    ans.BufferedRansEncoder = entropy_models.BufferedRansEncoder
    cxx = types.ModuleType("compressai._CXX")
    cxx.pmf_to_quantized_cdf = lambda pmf, precision: entropy_models.pmf_to_quantized_cdf(pmf, precision).tolist()
    models = types.ModuleType("compressai.models")
    models.__path__ = []
    for mod in (priors, spatiotemporalpriors, stem_roi, stem_utils):
        for name in mod.__all__:
            setattr(models, name, getattr(mod, name))
    table = {"compressai": root, "compressai.ans": ans, "compressai._CXX": cxx, "compressai.zoo": zoo,
             "compressai.models": models, "compressai.models.priors": priors, "compressai.models.utils": utils,
             "compressai.models.spatiotemporalpriors": spatiotemporalpriors, "compressai.models.stem_roi": stem_roi,
             "compressai.models.stem_utils": stem_utils, "compressai.layers": layers,
             "compressai.entropy_models": entropy_models, "compressai.ops": ops}
    for k, v in table.items():
        sys.modules[k] = v
        if "." in k:
            parent, leaf = k.rsplit(".", 1)
            setattr(table[parent], leaf, v)
    return root
