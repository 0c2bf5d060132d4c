"""Back-compat alias package (ref:neuttsair/__init__.py, ref:neuttsair/neutts.py:4-11)."""
from .neutts import NeuTTSAir  # noqa: F401

__all__ = ["NeuTTSAir"]
