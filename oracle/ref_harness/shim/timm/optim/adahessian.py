class Adahessian:
    pass
