"""Does the engine of a drop-in training run die when the run returns (by reference count), or later, in a pass of
the cycle collector?  Prints, after each run, whether the engine and its HIP graphs are still alive."""
import gc
import os
import sys
import weakref

sys.argv = [sys.argv[0], "0"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import diag_fast_plain as D  # noqa: E402  (runs zero repetitions, gives train())
from cycle_gan_rna_amd.engine import StepEngine  # noqa: E402

refs = []
_init = StepEngine.__init__


def init(self, *a, **k):
    _init(self, *a, **k)
    refs.append(weakref.ref(self))


StepEngine.__init__ = init
gc.disable()
for identity in (False, True):
    for mode in ("fast", "plain"):
        D.train(identity, mode)
        alive = refs[-1]() is not None
        print(f"identity={identity} {mode}: engine alive after the run returned: {alive}", flush=True)
        if alive:
            print("   referrers:", [type(r).__name__ for r in gc.get_referrers(refs[-1]())][:8], flush=True)
            n = gc.collect()
            print(f"   cycle collector freed {n} objects; alive now: {refs[-1]() is not None}", flush=True)
