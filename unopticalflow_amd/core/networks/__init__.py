"""Model registry of the flow stage (reference core/networks/__init__.py:5-9: ``get_model(mode)``)."""
from .model_flow_paper import Model_flow

_REGISTRY = {'flow': Model_flow}      # depth / pose modes of the upstream project are outside this package


def get_model(mode):
    """Class for a training mode; unknown modes raise ValueError with the reference's message."""
    model_cls = _REGISTRY.get(mode)
    if model_cls is None:
        raise ValueError('Mode {} not found.'.format(mode))
    return model_cls
