from .registry import _REGISTRY


def create_model(model_name, pretrained=False, **kwargs):
    # timm drops kwargs whose value is None before calling the factory
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _REGISTRY[model_name](pretrained=pretrained, **kwargs)
