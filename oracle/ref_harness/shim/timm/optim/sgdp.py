class SGDP:
    pass
