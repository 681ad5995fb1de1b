class AdamP:
    pass
