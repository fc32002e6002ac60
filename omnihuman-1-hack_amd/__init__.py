"""omnihuman-1-hack_amd — MI355X-native (gfx950) Wan2.1 DiT + 3D-causal-VAE path.

Drop-in for the hot path johndpope/OmniHuman-1-hack drives through
``wan.modules.model.WanModel`` / ``wan.modules.vae.WanVAE`` / ``wan.WanT2V``:
hand-written HIP kernels behind a C ABI (include/omh.h, csrc/), bound with
ctypes (_lib.py, ops.py), under the reference's own Python call surface
(wan/).  The directory name is not a Python identifier; import it with

    import importlib
    omh = importlib.import_module("omnihuman-1-hack_amd")
    WanModel = importlib.import_module("omnihuman-1-hack_amd.wan.modules.model").WanModel

or put this directory on ``sys.path`` and ``import wan`` exactly as the
reference's scripts do.  Importing the package loads
libomh.so and fails loudly when it is missing and cannot be built.
"""
from . import _lib  # noqa: F401  (loads libomh.so; raises if unavailable)
from . import ops  # noqa: F401

__version__ = "0.1.0"
