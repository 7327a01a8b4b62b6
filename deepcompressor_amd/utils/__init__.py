from . import common, config

__all__ = ["common", "config"]
