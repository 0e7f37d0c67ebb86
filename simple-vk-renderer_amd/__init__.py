"""MI355X-native replacement for the Vulkan geometry pass of imalexlee/simple-vk-renderer.

The product is the HIP library csrc/libsvr_hip.so behind the C ABI of include/svr.h
(VulkanEngine::draw_geometry, src/vk_engine.cpp:1357-1477).  This Python package is only the
ctypes binding, the synthetic scene generators of the BASELINE configs and the multi-GPU driver;
it never falls back to a CPU path: load_product_library() raises if the HIP library is missing.

The directory name contains a hyphen (the name the build contract asks for), so import it with
`__graft_entry__.load_package()` which registers it as module `simple_vk_renderer_amd`.
"""
import os

from . import abi, dist, glmath, scenes  # noqa: F401
from .abi import SvrLib, Renderer, SvrError  # noqa: F401

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PACKAGE_DIR)
PRODUCT_LIBRARY = os.path.join(PACKAGE_DIR, "csrc", "libsvr_hip.so")

_product = None


def load_product_library():
    """Load csrc/libsvr_hip.so.  Fails loudly when it has not been built: there is no fallback."""
    global _product
    if _product is None:
        if not os.path.exists(PRODUCT_LIBRARY):
            raise RuntimeError(
                f"{PRODUCT_LIBRARY} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  The product has no CPU fallback.")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64 and must be the one that
        # gets loaded (same SONAME as /opt/rocm's), or torch.cuda later finds "No HIP GPUs" and stream
        # handles could not be shared.  Import torch first whenever it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _product = SvrLib(PRODUCT_LIBRARY)
        if _product.backend != "hip-gfx950":
            raise RuntimeError(f"{PRODUCT_LIBRARY} reports backend {_product.backend!r}, expected 'hip-gfx950'")
    return _product
