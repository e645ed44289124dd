"""Debugging aid: make a crash leave evidence.  install() switches on Python's faulthandler (Python frames of every
thread) and, when tools/libsegvtrace.so has been built (gcc -O1 -g -fPIC -shared -o tools/libsegvtrace.so
tools/segv_trace.c), a handler that prints the native stack of the faulting thread first (module + offset per frame).

    python tools/faultdbg.py bench.py --steps 5        runs a script under both handlers
"""
import ctypes
import faulthandler
import os
import runpy
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def install():
    faulthandler.enable(all_threads=True)
    so = os.path.join(_HERE, "libsegvtrace.so")
    if os.path.exists(so):
        ctypes.CDLL(so).segv_trace_install()
        return True
    return False


if __name__ == "__main__":
    install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")
