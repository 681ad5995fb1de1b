class RMSpropTF:
    pass
