class NovoGrad:
    pass
