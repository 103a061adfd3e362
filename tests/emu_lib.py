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
        import threading
        import build_emu
        from deeplocalproteindocking_amd._lib import DlpdLib

        class SerialisedEmu(DlpdLib):
            """The emulator keeps its "wavefront" state and LDS in globals: one kernel at a time.  (ctypes releases the
            GIL during a call, so two host threads -- a sweep preparing the next target -- would otherwise be inside.)"""
            _one_kernel = threading.Lock()

            def call(self, name, *args):
                with self._one_kernel:
                    return DlpdLib.call(self, name, *args)
        _lib = SerialisedEmu(build_emu.build())
    return _lib
