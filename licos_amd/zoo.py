"""``image_models`` mapping with CompressAI's call signature
(``image_models[name](quality=int, pretrained=bool)``), the factory
/root/reference/licos/model_utils.py:19 and licos/train.py:111-113 call.
Topology per quality follows CompressAI ``zoo/image.py``: q1-5 -> (N, M) = (128, 192),
q6-8 -> (192, 320)."""
from .models import FactorizedPrior, FactorizedPriorReLU, ScaleHyperprior

_CFGS = {q: ((128, 192) if q <= 5 else (192, 320)) for q in range(1, 9)}


def _load(cls, name, quality, pretrained, **kwargs):
    if quality not in _CFGS:
        raise ValueError(f'Invalid quality "{quality}", should be between (1, 8)')
    if pretrained:
        # the reference downloads weights from a URL here; this build has no network access
        raise RuntimeError(f"licos_amd: pretrained weights for {name} are not bundled (no network); "
                           "load a CompressAI/LICOS checkpoint with load_state_dict instead")
    n, m = _CFGS[quality]
    return cls(n, m, **kwargs)


def bmshj2018_factorized(quality, metric="mse", pretrained=False, progress=True, **kwargs):
    return _load(FactorizedPrior, "bmshj2018-factorized", quality, pretrained, **kwargs)


def bmshj2018_factorized_relu(quality, metric="mse", pretrained=False, progress=True, **kwargs):
    return _load(FactorizedPriorReLU, "bmshj2018-factorized-relu", quality, pretrained, **kwargs)


def bmshj2018_hyperprior(quality, metric="mse", pretrained=False, progress=True, **kwargs):
    return _load(ScaleHyperprior, "bmshj2018-hyperprior", quality, pretrained, **kwargs)


image_models = {
    "bmshj2018-hyperprior": bmshj2018_hyperprior,
    "bmshj2018-factorized": bmshj2018_factorized,
    "bmshj2018-factorized-relu": bmshj2018_factorized_relu,
}
