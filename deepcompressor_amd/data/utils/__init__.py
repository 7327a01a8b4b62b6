from . import dtype as DtypeUtils
from . import scale as ScaleUtils
from . import shape as ShapeUtils

__all__ = ["DtypeUtils", "ScaleUtils", "ShapeUtils"]
