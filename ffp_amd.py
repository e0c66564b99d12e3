"""Importable alias for the package directory `face-detection-with-yolov11-sahi-and-real-esrgan_amd`
(its mandated name contains hyphens, so `import` statements cannot spell it)."""
import importlib
import sys

_pkg = importlib.import_module("face-detection-with-yolov11-sahi-and-real-esrgan_amd")
sys.modules[__name__] = _pkg
