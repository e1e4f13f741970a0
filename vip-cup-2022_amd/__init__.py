"""vipcup_amd — MI355X-native scoring path for awsaf49/vip-cup-2022 (main.py:58-149).

Host side: Python mirrors of the reference's model constructors and dataset/ensemble functions.
Device side: libvipcup_hip.so (hand-written HIP for gfx950) behind the C ABI of include/vipcup_hip.h.
There is no CPU fallback: every op raises if the HIP library is missing.
"""
__version__ = "0.1.0"
