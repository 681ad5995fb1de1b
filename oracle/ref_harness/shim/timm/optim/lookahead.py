class Lookahead:
    pass
