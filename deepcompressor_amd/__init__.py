"""MI355X-native calibration / quantization engine behind deepcompressor's ``Quantizer`` / ``calib`` API."""

__version__ = "0.1.0"
