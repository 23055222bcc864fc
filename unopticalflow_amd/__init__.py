"""unopticalflow_amd -- MI355X-native hot path of UnOpticalFlow's ``--mode flow``.

Host side mirrors the reference's ``core/networks`` operator surface (``get_model``,
``Model_flow``, ``FeaturePyramid``, ``PWC_tf``, ``conv``, ``deconv``, ``warp_flow``, ``SSIM``);
the cost volume, warp, SSIM / occlusion / loss reductions run as hand-written gfx950 kernels
behind the C ABI of ``include/unflow_hip.h`` (``libunflow_hip.so``); convolutions run on MFMA
through PyTorch-ROCm.  No CPU fallback.
"""
from .core.networks import get_model, Model_flow                               # noqa: F401
from .core.networks.structures import FeaturePyramid, PWC_tf, conv, deconv, warp_flow   # noqa: F401
from .core.networks.pytorch_ssim import SSIM                                   # noqa: F401
from .core.config import generate_loss_weights_dict                            # noqa: F401

__version__ = '0.1.0'
