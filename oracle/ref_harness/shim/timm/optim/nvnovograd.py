class NvNovoGrad:
    pass
