from .... import ops


def SSIM(x, y):
    """3x3 mean-filter SSIM map (reference core/networks/pytorch_ssim/ssim.py:4-20), one HIP
    sliding-window kernel instead of five AvgPool2d + ~15 elementwise launches.

    Forward-only: training goes through ``ops.ssim_loss`` (the fused loss + its own backward),
    which is what ``Model_flow.compute_loss_ssim`` calls.
    """
    return ops.ssim_map(x, y)
