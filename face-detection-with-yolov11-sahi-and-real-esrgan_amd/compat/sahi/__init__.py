"""Subset of the `sahi` package surface the reference imports, implemented over libffp.so."""
__version__ = "0.11.34+ffp"
