class Nadam:
    pass
