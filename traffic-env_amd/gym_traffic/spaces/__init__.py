from .gspace import GSpace  # noqa: F401
