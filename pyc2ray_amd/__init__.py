"""pyc2ray_amd -- MI355X-native implementation of pyC2Ray's ASORA raytracing path.

Importing the package does not load the HIP library; ``pyc2ray_amd.libasora`` does, and raises
``ImportError`` when ``lib/libasora_hip.so`` has not been built (there is no CPU fallback)."""
__all__ = ["radiation", "sourceutils"]
