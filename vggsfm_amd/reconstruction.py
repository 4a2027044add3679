"""The reconstruction object the drop-in returns: :class:`vggsfm_amd.pycolmap_compat.Reconstruction` (the pycolmap
object surface over flat arrays, COLMAP ``.bin`` reader / writer included).  This module keeps the old import path."""
from .pycolmap_compat import CAMERA_MODEL_IDS, Reconstruction, rotmat_to_qvec  # noqa: F401
