"""Import harness for the read-only reference tree (TEST INFRASTRUCTURE ONLY).

The reference (``/root/reference``, facebookresearch/vggsfm v2.0.0) imports pycolmap,
pyceres, kornia, cv2, hydra ... at module level; none are installed here.  This module
installs a ``sys.meta_path`` finder that fabricates empty packages for those prefixes so
that the *torch half* of the hot path (``vggsfm/utils/triangulation.py``,
``triangulation_helpers.py``, ``distortion.py``, ``two_view_geo/utils.py``) can be
imported and run on the CPU.  It is used ONLY by ``oracle/gen_golden.py`` (run in the
build container, where ``/root/reference`` exists) to produce ``tests/golden/*.npz``.
Nothing in ``vggsfm_amd/`` may import it, and nothing that runs on the GPU box does:
``/root/reference`` does not exist there.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("VGGSFM_REFERENCE_ROOT", "/root/reference")

_STUB_PREFIXES = (
    "pycolmap", "pyceres", "poselib", "cv2", "kornia", "hydra", "omegaconf", "lightglue",
    "visdom", "pytorch3d", "torchvision", "imageio", "trimesh", "gradio", "h5py", "flow_vis",
)


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        value = mock.MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, value)
        return value


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        return None


_installed = False


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vggsfm"))


def install() -> None:
    """Make ``import vggsfm...`` resolve to the reference tree with third-party stubs."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def load():
    """Return (triangulation, triangulation_helpers, distortion) reference modules."""
    install()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from vggsfm.utils import triangulation, triangulation_helpers, distortion  # noqa
    return triangulation, triangulation_helpers, distortion
