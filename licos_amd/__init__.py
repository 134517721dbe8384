"""licos_amd - MI355X (gfx950) native implementation of the LICOS learned-image-compression
hot path: bmshj2018-factorized analysis/synthesis transforms, entropy bottleneck and the
bit-exact rANS stream, behind CompressAI's nn.Module surface.  See DESIGN.md."""
import os as _os

# The hyperprior codec keeps one serial coder launch per pipeline chunk in flight, each on a stream of its own; HIP maps
# streams onto 4 hardware queues by default, where the fifth stream's launch queues up behind another's ~50 ms.  Read by
# the HIP runtime when it initialises (first device call), so this must come before anything touches the GPU.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .entropy_models import EntropyBottleneck, GaussianConditional  # noqa: F401
from .layers import GDN  # noqa: F401
from . import checkpoint, metrics  # noqa: F401
from .losses import RateDistortionLoss  # noqa: F401
from .model_utils import get_model  # noqa: F401
from .models import FactorizedPrior, FactorizedPriorReLU, ScaleHyperprior  # noqa: F401
from .optimizers import net_aux_optimizer  # noqa: F401
from .zoo import image_models  # noqa: F401

__version__ = "0.1.0"
