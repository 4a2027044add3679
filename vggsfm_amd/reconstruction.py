"""Tensor-backed stand-in for the ``pycolmap.Reconstruction`` the reference returns.

The reference builds a pycolmap object with an O(S*P) Python loop before EVERY bundle adjustment
(vggsfm/utils/tensor_to_pycolmap.py:76-158).  The device path never needs it for solving; this facade
is only materialised once, at the end, for the callers that read or export the model
(vggsfm/runners/runner.py:555-611, 887-911, 1009-1054).  It keeps flat numpy arrays and writes the
COLMAP binary model directly (format as read by the reference's own reader,
vggsfm/datasets/imc_helper.py:127-159, 238-286, 380-416).
"""
import os
import struct

import numpy as np

CAMERA_MODEL_IDS = {"SIMPLE_PINHOLE": 0, "SIMPLE_RADIAL": 2}


def rotmat_to_qvec(R):
    """COLMAP qvec (w,x,y,z) of a rotation matrix (Eigen conversion)."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()        # x,y,z,w
    q = q[[3, 0, 1, 2]]
    return q if q[0] >= 0 else -q


class Reconstruction:
    def __init__(self, points3D, extrinsics, intrinsics, tracks, masks, image_size, shared_camera=False,
                 camera_type="SIMPLE_PINHOLE", extra_params=None, colors=None):
        """points3D (P,3), extrinsics (S,3,4), intrinsics (S,3,3), tracks (S,P,2), masks (S,P) bool -- numpy."""
        self.points3D_xyz = np.asarray(points3D, np.float64)
        self.extrinsics = np.asarray(extrinsics, np.float64)
        self.intrinsics = np.asarray(intrinsics, np.float64)
        self.tracks = np.asarray(tracks, np.float64)
        self.masks = np.asarray(masks, bool)
        self.image_size = np.asarray(image_size).reshape(-1)[:2]
        self.shared_camera = bool(shared_camera)
        self.camera_type = camera_type
        self.extra_params = None if extra_params is None else np.asarray(extra_params, np.float64)
        P = len(self.points3D_xyz)
        self.colors = np.zeros((P, 3), np.uint8) if colors is None else np.asarray(colors, np.uint8)

    # ---- the small read surface the runners use
    def num_points3D(self):
        return len(self.points3D_xyz)

    def num_images(self):
        return len(self.extrinsics)

    def point3D_ids(self):
        return list(range(1, self.num_points3D() + 1))          # 1-based like tensor_to_pycolmap.py:127

    def camera_params(self, s):
        K = self.intrinsics[0 if self.shared_camera else s]
        p = [K[0, 0], K[0, 2], K[1, 2]]
        if self.camera_type == "SIMPLE_RADIAL":
            p.append(self.extra_params[0 if self.shared_camera else s, 0])
        return np.array(p, np.float64)

    # ---- COLMAP binary model
    def write(self, path):
        os.makedirs(path, exist_ok=True)
        S, P = self.masks.shape
        cam_ids = [0] if self.shared_camera else list(range(S))
        with open(os.path.join(path, "cameras.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(cam_ids)))
            for cid in cam_ids:
                prm = self.camera_params(cid)
                f.write(struct.pack("<iiQQ", cid, CAMERA_MODEL_IDS[self.camera_type], int(self.image_size[0]),
                                    int(self.image_size[1])))
                f.write(struct.pack(f"<{len(prm)}d", *prm))
        # point2D index of observation (s,p) inside image s = rank among the masked points of that image
        p2d_idx = np.cumsum(self.masks, axis=1) - 1
        with open(os.path.join(path, "images.bin"), "wb") as f:
            f.write(struct.pack("<Q", S))
            for s in range(S):
                q = rotmat_to_qvec(self.extrinsics[s, :, :3])
                t = self.extrinsics[s, :, 3]
                f.write(struct.pack("<i4d3di", s, *q, *t, 0 if self.shared_camera else s))
                f.write(f"image_{s}".encode() + b"\x00")
                pids = np.nonzero(self.masks[s])[0]
                f.write(struct.pack("<Q", len(pids)))
                rec = np.empty(len(pids), dtype=[("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
                rec["x"], rec["y"], rec["id"] = self.tracks[s, pids, 0], self.tracks[s, pids, 1], pids + 1
                f.write(rec.tobytes())
        with open(os.path.join(path, "points3D.bin"), "wb") as f:
            f.write(struct.pack("<Q", P))
            for p in range(P):
                frames = np.nonzero(self.masks[:, p])[0]
                f.write(struct.pack("<Q3d3Bd", p + 1, *self.points3D_xyz[p], *self.colors[p].tolist(), 0.0))
                f.write(struct.pack("<Q", len(frames)))
                tr = np.empty(len(frames), dtype=[("im", "<i4"), ("pt", "<i4")])
                tr["im"], tr["pt"] = frames, p2d_idx[frames, p]
                f.write(tr.tobytes())
