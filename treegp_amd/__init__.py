"""treegp_amd -- MI355X-native GP interpolation hot path with treegp's Python API.

Same public names as ``treegp/__init__.py:7-36`` of the reference for everything on the hot
path (``two_pcf`` and ``log_likelihood`` are the classes, shadowing their modules, as there).
All numerical work happens in ``csrc/libtgp.so`` (hand-written HIP for gfx950) behind the C-ABI
of ``include/tgp.h``; importing the package needs no GPU, computing does.
"""
__version__ = "0.1.0"
__version_info__ = tuple(map(int, __version__.split(".")))

from .gp_interp import GPInterpolation
from .kernels import AnisotropicRBF, VonKarman, AnisotropicVonKarman, eval_kernel, kernel_to_spec
from .two_pcf import two_pcf  # noqa: F401  (class shadows the module, as in the reference)
from .log_likelihood import log_likelihood  # noqa: F401
from .meanify import meanify  # noqa: F401
from .utils import comp_eb, comp_eb_treecorr

__all__ = ["__version__", "__version_info__", "GPInterpolation", "two_pcf", "log_likelihood", "AnisotropicRBF",
           "VonKarman", "AnisotropicVonKarman", "eval_kernel", "kernel_to_spec", "meanify", "comp_eb", "comp_eb_treecorr"]
