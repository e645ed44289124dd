"""MI355X-native CycleGAN / cycle-cGAN training step for single-cell expression
(drop-in for the hot path of guyshani/cycle-GAN-RNA's GAN_functions.py).

Layout:
  csrc/          HIP kernels (gfx950) and the C ABI declared in include/cgrna.h
  _lib.py        ctypes loader of lib/libcgrna.so
  ops.py         tensor-level wrappers, one kernel launch per method
  engine.py      the explicit forward/backward/update of one training iteration
  GAN_functions.py  the reference's Python API on top of the engine
"""
__version__ = "0.1.0"
