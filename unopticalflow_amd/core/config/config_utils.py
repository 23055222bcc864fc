"""Loss weighting of the flow stage (reference core/config/config_utils.py:3-9)."""

_LOSS_KEYS = ('loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis')


def generate_loss_weights_dict(cfg):
    """{loss name: weight}: the photometric term gets what SSIM leaves (w_pixel = 1 - w_ssim)."""
    weights = (1 - cfg.w_ssim, cfg.w_ssim, cfg.w_flow_smooth, cfg.w_flow_consis)
    return dict(zip(_LOSS_KEYS, weights))
