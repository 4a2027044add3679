"""Drop-in for ``vggsfm.utils.tensor_to_pycolmap`` (vggsfm/utils/tensor_to_pycolmap.py:16-214) without pycolmap.

``batch_matrix_to_pycolmap`` returns a :class:`vggsfm_amd.pycolmap_compat.Reconstruction` -- the pycolmap object
surface (``images`` / ``cameras`` / ``points3D`` mappings, ``add_point3D``, ``deregister_image``, ``normalize``,
``write`` ...) over flat arrays -- built with S numpy operations instead of the reference's O(S*P) Python loop;
``pycolmap_to_batch_matrix`` reads one back (any object with the pycolmap surface works).
The selection rules of the reference are kept: a track enters when it has >= 2 masked observations; observations of
points with a coordinate >= max_points3D_val are dropped; point3D ids are 1-based over the kept tracks; one camera
per frame, or a single one (frame 0's) when `shared_camera`.
The solver entry that consumes the same arguments directly is ``vggsfm_amd.ba.bundle_adjustment``.
"""
import numpy as np
import torch

from ..pycolmap_compat import Reconstruction


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
    if torch.is_tensor(tracks) and tracks.is_cuda and torch.is_tensor(masks) and torch.is_tensor(points3d):
        # (mixed placement is accepted, as by the reference, which moves everything to the CPU first: the selection runs
        #  where the tracks are -- ADVICE r4)
        masks, points3d = masks.to(tracks.device), points3d.to(tracks.device)
        # the selection on the device, only the kept observations travel (c3: 5 M rows instead of the 200 x 100 k grid and
        # 200 host-side gathers -- 0.24 s of Triangulator.forward's 1.26 s in round 3)
        m = masks.bool()
        valid_idx = torch.nonzero(m.sum(0) >= 2).squeeze(1)
        pts = points3d.to(torch.float64)[valid_idx]
        m2 = m[:, valid_idx] & (pts < max_points3D_val).all(-1)[None]
        f, j = torch.nonzero(m2, as_tuple=True)                       # frame-major, tracks ascending inside a frame
        xy = tracks[f, valid_idx[j]].to(torch.float64)
        counts = torch.bincount(f, minlength=N)
        return Reconstruction.from_frame_lists(_np(pts), _np(valid_idx), _np(extrinsics), _np(intrinsics), _np(xy), _np(j + 1),
                                               _np(counts), _np(image_size), shared_camera, camera_type, _np(extra_params))
    return Reconstruction.from_arrays(_np(points3d), _np(extrinsics), _np(intrinsics), _np(tracks), _np(masks),
                                      _np(image_size), max_points3D_val, shared_camera, camera_type, _np(extra_params))


def pycolmap_to_batch_matrix(reconstruction, device="cuda", camera_type="SIMPLE_PINHOLE"):
    """Reconstruction -> (points3D (max id,3), extrinsics (N,3,4), intrinsics (N,3,3), extra_params (N,1) | None)
    (reference lines 163-214): rows of deleted points stay zero; image i of 0..N-1 gives row i."""
    num_images = len(reconstruction.images)
    if isinstance(reconstruction, Reconstruction):
        n = max(reconstruction.point3D_ids())
        pts = np.where(reconstruction._alive[:n, None], reconstruction._xyz[:n], 0.0)
    else:
        pts = np.zeros((max(reconstruction.point3D_ids()), 3))
        for pid in reconstruction.points3D:
            pts[pid - 1] = reconstruction.points3D[pid].xyz
    ext, K, extra = [], [], []
    for i in range(num_images):
        img = reconstruction.images[i]
        cam = reconstruction.cameras[img.camera_id]
        ext.append(img.cam_from_world.matrix())
        K.append(cam.calibration_matrix())
        extra.append(cam.params[-1])
    extra_params = None
    if camera_type == "SIMPLE_RADIAL":
        extra_params = torch.from_numpy(np.stack(extra)).to(device)[:, None]
    return (torch.from_numpy(pts).to(device), torch.from_numpy(np.stack(ext)).to(device),
            torch.from_numpy(np.stack(K)).to(device), extra_params)
