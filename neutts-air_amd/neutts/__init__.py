"""MI355X-native NeuTTS hot path behind the reference's class surface (ref:neutts/__init__.py).

`NeuTTS` is resolved lazily so that the thin native binding (`neutts._hip`) can be imported on its own
(bench.py, kernel tests) without pulling the optional text/audio front-end imports.
"""
__all__ = ["NeuTTS"]


def __getattr__(name):
    if name == "NeuTTS":
        from .neutts import NeuTTS
        return NeuTTS
    raise AttributeError(name)
