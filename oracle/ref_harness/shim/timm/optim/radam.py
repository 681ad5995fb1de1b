class RAdam:
    pass
