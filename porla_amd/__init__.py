"""porla_amd -- host-side Python mirror of Porla's commitment plug-in boundary over the MI355X engine.

The product is porla_amd/libmultiexp.so (C ABI: include/libmultiexp.h + include/porla_gpu.h, HIP kernels
for gfx950).  This package only loads it with ctypes and mirrors the reference's C++ wrapper names
(porla/Utils/utils.h:235-305) so that tests read like calls the reference's Server/Client make.
There is no CPU fallback: if the shared object is missing the import of `porla_amd.lib` raises.
"""
from .loader import lib, lib_path, load  # noqa: F401
