class InvalidFrame(Exception):
    pass
