"""ref:neuttsair/neutts.py:1-11 -- NeuTTSAir is NeuTTS under its former name."""
from neutts.neutts import NeuTTS


class NeuTTSAir(NeuTTS):
    """Subclass alias kept for back-compat; inherits everything (ref:neuttsair/neutts.py:4-11)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
