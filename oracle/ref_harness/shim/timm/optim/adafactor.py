class Adafactor:
    pass
