"""Model registry with the `timm.models.create_model` calling convention the reference driver uses
(run_mae_pretraining_moco.py:278-294)."""
_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def create_model(model_name, pretrained=False, **kwargs):
    from . import modeling_pretrain_moco_mim_ori  # noqa: F401  (registers the factories)
    if model_name not in _REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name})")
    kwargs = {k: v for k, v in kwargs.items() if v is not None}      # timm drops None-valued kwargs
    return _REGISTRY[model_name](pretrained=pretrained, **kwargs)
