"""vlgp_amd -- MI355X-native variational-EM engine for vLGP: a drop-in for the
``vlgp.fit`` hot path of catniplab/vlgp (hand-written HIP kernels behind a C ABI,
NumPy host)."""
from .api import fit, sample_posterior, transform  # noqa: F401
from .util import load, save  # noqa: F401
from .engine import (DeviceTrials, Engine, VlgpError, constrain_latent, constrain_loading,  # noqa: F401
                     estep, hstep, infer, make_cholesky, mstep, update_v, update_w, vem)
from .preprocess import get_config, get_params  # noqa: F401

__all__ = ["fit", "transform", "Engine", "DeviceTrials", "VlgpError"]
