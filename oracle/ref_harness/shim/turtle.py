class Turtle:  # the reference engine imports this name by accident (needs tkinter otherwise)
    pass
