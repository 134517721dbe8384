"""licos_amd - MI355X (gfx950) native implementation of the LICOS learned-image-compression
hot path: bmshj2018-factorized analysis/synthesis transforms, entropy bottleneck and the
bit-exact rANS stream, behind CompressAI's nn.Module surface.  See DESIGN.md."""
from .entropy_models import EntropyBottleneck, GaussianConditional  # noqa: F401
from .layers import GDN  # noqa: F401
from . import checkpoint, metrics  # noqa: F401
from .losses import RateDistortionLoss  # noqa: F401
from .model_utils import get_model  # noqa: F401
from .models import FactorizedPrior, FactorizedPriorReLU, ScaleHyperprior  # noqa: F401
from .optimizers import net_aux_optimizer  # noqa: F401
from .zoo import image_models  # noqa: F401

__version__ = "0.1.0"
