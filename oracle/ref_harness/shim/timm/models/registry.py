_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn
