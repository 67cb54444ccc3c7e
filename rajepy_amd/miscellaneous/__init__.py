from . import functions  # noqa: F401
