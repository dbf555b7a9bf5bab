"""Name -> constructor registry (compressai/zoo/__init__.py:17-24, compressai/zoo/image.py:131-215) for the
one architecture the STEM scripts instantiate: `models["mbt2018"](quality=4)` (stem/trainSTEM.py:113)."""
from .models.priors import JointAutoregressiveHierarchicalPriors

cfgs = {"mbt2018": {1: (192, 192), 2: (192, 192), 3: (192, 192), 4: (192, 192),
                    5: (192, 320), 6: (192, 320), 7: (192, 320), 8: (192, 320)}}


def mbt2018(quality, metric="mse", pretrained=False, progress=True, **kwargs):
    if metric not in ("mse",):
        raise ValueError(f'Invalid metric "{metric}"')
    if quality not in cfgs["mbt2018"]:
        raise ValueError(f'Invalid quality "{quality}", should be between (1, 8)')
    if pretrained:
        raise RuntimeError("pretrained weights are downloaded from S3 by the reference (zoo/image.py:46); no network here")
    return JointAutoregressiveHierarchicalPriors(*cfgs["mbt2018"][quality], **kwargs)


models = {"mbt2018": mbt2018}
