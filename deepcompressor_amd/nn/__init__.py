from . import patch

__all__ = ["patch"]
