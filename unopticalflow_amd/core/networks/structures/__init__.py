from .feature_pyramid import FeaturePyramid          # noqa: F401
from .pwc_tf import PWC_tf                           # noqa: F401
from .net_utils import conv, deconv, warp_flow       # noqa: F401
