"""Mirror of ``pyc2ray/load_extensions.py:12-54`` for the one extension this package has.

The reference returns ``None`` when the library is missing and lets callers fall back to the CPU;
here a missing library is an error (no fallback exists on this path)."""
from __future__ import annotations


def load_asora():
    """Return the ``libasora`` module (raises ImportError when libasora_hip.so is not built)."""
    from . import libasora
    return libasora
