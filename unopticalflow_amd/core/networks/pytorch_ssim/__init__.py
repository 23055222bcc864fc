from .ssim import SSIM   # noqa: F401
