from . import seeding  # noqa: F401


def colorize(s, *a, **k):
    return s
