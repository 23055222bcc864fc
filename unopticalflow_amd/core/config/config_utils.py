def generate_loss_weights_dict(cfg):
    """Loss weights of the flow stage (reference core/config/config_utils.py:3-9)."""
    return {
        'loss_pixel': 1 - cfg.w_ssim,
        'loss_ssim': cfg.w_ssim,
        'loss_flow_smooth': cfg.w_flow_smooth,
        'loss_flow_consis': cfg.w_flow_consis,
    }
