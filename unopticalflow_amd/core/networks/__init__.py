from .model_flow_paper import Model_flow


def get_model(mode):
    """reference core/networks/__init__.py:5-9"""
    if mode == 'flow':
        return Model_flow
    else:
        raise ValueError('Mode {} not found.'.format(mode))
