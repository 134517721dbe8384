"""Mirror of /root/reference/licos/model_utils.py:6-49 over this package's zoo: same
name, arguments, channel surgery and error behaviour, so the reference file works with
``from licos_amd.zoo import image_models`` / ``from licos_amd.entropy_models import
EntropyBottleneck`` substituted for its two CompressAI imports."""
from torch.nn import Conv2d, ConvTranspose2d

from .entropy_models import EntropyBottleneck
from .zoo import image_models

_RAW_MODELS = ("bmshj2018-factorized", "bmshj2018-factorized-relu", "bmshj2018-hyperprior")


def get_model(model, pretrained, in_channels=3, quality=1):
    if model in _RAW_MODELS and model not in image_models:
        raise NotImplementedError(f"licos_amd: {model} is not built yet")
    if model not in image_models and model not in _RAW_MODELS:
        # the reference indexes the zoo first (KeyError for unknown names); keep that
        raise KeyError(model)
    net = image_models[model](quality=quality, pretrained=pretrained)
    if model not in _RAW_MODELS:
        raise ValueError("model: " + model + " not supported for raw data.")
    net.entropy_bottleneck = EntropyBottleneck(
        channels=net.entropy_bottleneck.channels,
        filters=(in_channels, in_channels, 3, 3),
    )
    first = net.g_a[0]
    net.g_a[0] = Conv2d(
        in_channels=in_channels,
        out_channels=first.out_channels,
        kernel_size=(first.weight.shape[2], first.weight.shape[3]),
        stride=first.stride,
        padding=first.padding,
    )
    last = net.g_s[6]
    net.g_s[6] = ConvTranspose2d(
        in_channels=last.in_channels,
        out_channels=in_channels,
        kernel_size=(last.weight.shape[2], last.weight.shape[3]),
        stride=last.stride,
        padding=last.padding,
        output_padding=last.output_padding,
    )
    return net
