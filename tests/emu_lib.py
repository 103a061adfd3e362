"""Load the CPU-emulated kernel library (tests/emu) behind the product ctypes binding."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
sys.path.insert(0, os.path.join(HERE, "emu"))

_lib = None


def emu_lib():
    global _lib
    if _lib is None:
        import build_emu
        from deeplocalproteindocking_amd._lib import DlpdLib
        _lib = DlpdLib(build_emu.build())
    return _lib
