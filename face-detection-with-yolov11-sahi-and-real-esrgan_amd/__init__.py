"""MI355X-native sliced-inference face pipeline (SAHI slice -> YOLO11-pose -> NMS -> SAHI merge -> Real-ESRGAN)."""
__version__ = "0.1.0"
