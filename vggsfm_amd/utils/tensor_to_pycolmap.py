"""Drop-in for ``vggsfm.utils.tensor_to_pycolmap`` (vggsfm/utils/tensor_to_pycolmap.py:16-214) without pycolmap.

``batch_matrix_to_pycolmap`` returns the tensor-backed :class:`vggsfm_amd.reconstruction.Reconstruction` facade
instead of a ``pycolmap.Reconstruction`` built by an O(S*P) Python loop; ``pycolmap_to_batch_matrix`` reads it back.
The selection rules of the reference are kept: a track enters when it has >= 2 masked observations; observations of
points with a coordinate >= max_points3D_val are dropped; point3D ids are 1-based over the kept tracks; one camera
per frame, or a single one (frame 0's) when `shared_camera`.
The solver entry that consumes the same arguments directly is ``vggsfm_amd.ba.bundle_adjustment``.
"""
import numpy as np
import torch

from ..reconstruction import Reconstruction


def _np(x):
    return None if x is None else (x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x))


def batch_matrix_to_pycolmap(points3d, extrinsics, intrinsics, tracks, masks, image_size, max_points3D_val=3000,
                             shared_camera=False, camera_type="SIMPLE_PINHOLE", extra_params=None):
    """points3d (P,3), extrinsics (N,3,4), intrinsics (N,3,3), tracks (N,P,2), masks (N,P), image_size (2,)
    -> Reconstruction over the tracks with >= 2 inliers (reference lines 44-158)."""
    if camera_type not in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL"):
        raise ValueError(f"Camera type {camera_type} is not supported yet")
    N, P, _ = tracks.shape
    assert len(extrinsics) == N and len(intrinsics) == N and len(points3d) == P and image_size.shape[0] == 2
    pts, ext, K = _np(points3d), _np(extrinsics), _np(intrinsics)
    trk, msk, xp = _np(tracks), _np(masks).astype(bool), _np(extra_params)
    valid_idx = np.nonzero(msk.sum(0) >= 2)[0]
    p = pts[valid_idx]
    m = msk[:, valid_idx] & (p < max_points3D_val).all(-1)[None]
    if shared_camera:                                  # the single camera carries frame 0's parameters
        K = np.repeat(K[0:1], N, 0)
        xp = None if xp is None else np.repeat(xp[0:1], N, 0)
    rec = Reconstruction(p, ext, K, trk[:, valid_idx], m, _np(image_size), shared_camera=shared_camera,
                         camera_type=camera_type, extra_params=xp if camera_type == "SIMPLE_RADIAL" else None)
    rec.valid_idx = valid_idx
    return rec


def pycolmap_to_batch_matrix(reconstruction, device="cuda", camera_type="SIMPLE_PINHOLE"):
    """Reconstruction -> (points3D (P',3), extrinsics (N,3,4), intrinsics (N,3,3), extra_params (N,1) | None)
    (reference lines 163-214)."""
    points3D = torch.from_numpy(np.ascontiguousarray(reconstruction.points3D_xyz)).to(device)
    extrinsics = torch.from_numpy(np.ascontiguousarray(reconstruction.extrinsics)).to(device)
    intrinsics = torch.from_numpy(np.ascontiguousarray(reconstruction.intrinsics)).to(device)
    extra_params = None
    if camera_type == "SIMPLE_RADIAL":
        xp = reconstruction.extra_params
        extra_params = torch.from_numpy(np.ascontiguousarray(xp[:, :1])).to(device)
    return points3D, extrinsics, intrinsics, extra_params
