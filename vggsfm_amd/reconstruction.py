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
        S = len(self.extrinsics)
        self.image_names = [f"image_{s}" for s in range(S)]
        self.registered = np.ones(S, bool)
        ncam = 1 if self.shared_camera else S
        self.camera_sizes = np.tile(np.asarray(self.image_size, np.int64)[None], (ncam, 1))       # (width, height) per camera
        self.extra_points_xyz = np.zeros((0, 3), np.float64)      # points without a track (dense extra points)
        self.extra_points_rgb = np.zeros((0, 3), np.uint8)

    # ---- the small read surface the runners use
    def num_points3D(self):
        return len(self.points3D_xyz) + len(self.extra_points_xyz)

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

    # ---- the edits the runner applies after the solve (vggsfm/runners/runner.py:546-591, 1009-1054)
    def add_points3D(self, xyz, rgb):
        """``reconstruction.add_point3D(xyz, pycolmap.Track(), rgb)`` for many points: no observations (runner.py:549-554)."""
        self.extra_points_xyz = np.concatenate([self.extra_points_xyz, np.asarray(xyz, np.float64).reshape(-1, 3)])
        self.extra_points_rgb = np.concatenate([self.extra_points_rgb, np.asarray(rgb, np.uint8).reshape(-1, 3)])

    def deregister_image(self, image_id):
        """runner.py:567-569: the image stays in the model without a pose; its observations are dropped on write."""
        self.registered[int(image_id)] = False

    def rename_and_rescale(self, image_paths, crop_params, img_size, shift_point2d_to_original_res=False):
        """``rename_colmap_recons_and_rescale_camera`` (runner.py:1009-1054): images get their file names; cameras go
        back to the original resolution -- focal x max(real size) / img_size, principal point = real size // 2, width /
        height = real size (one camera only when shared); optionally the 2D points too:
        (xy - |crop top-left|) x that ratio.  crop_params (1,S,>=4): [..., :2] real (w, h), [..., -4:-2] top-left."""
        cp = np.asarray(crop_params, np.float64)[0]
        S = len(self.extrinsics)
        self.image_names = [str(image_paths[s]) for s in range(S)]
        ncam = 1 if self.shared_camera else S
        K = self.intrinsics.copy()
        ratio_of_image = np.zeros(S)
        ratio = None
        for s in range(S):
            if s < ncam:
                real = cp[s, :2]
                ratio = real.max() / float(img_size)
                cams = range(S) if self.shared_camera else [s]
                for c in cams:
                    K[c, 0, 0] = ratio * self.intrinsics[c, 0, 0]
                    K[c, 1, 1] = ratio * self.intrinsics[c, 1, 1]
                    K[c, 0, 2], K[c, 1, 2] = real[0] // 2, real[1] // 2
                self.camera_sizes[s] = real.astype(np.int64)
            ratio_of_image[s] = ratio            # (shared camera: the ratio of image 0 serves every image, as in the reference)
            if shift_point2d_to_original_res:
                self.tracks[s] = (self.tracks[s] - np.abs(cp[s, -4:-2])[None]) * ratio_of_image[s]
        self.intrinsics = K
        return self

    # ---- COLMAP binary model
    def write(self, path):
        os.makedirs(path, exist_ok=True)
        S, P = self.masks.shape
        cam_ids = [0] if self.shared_camera else list(range(S))
        with open(os.path.join(path, "cameras.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(cam_ids)))
            for cid in cam_ids:
                prm = self.camera_params(cid)
                f.write(struct.pack("<iiQQ", cid, CAMERA_MODEL_IDS[self.camera_type], int(self.camera_sizes[cid][0]),
                                    int(self.camera_sizes[cid][1])))
                f.write(struct.pack(f"<{len(prm)}d", *prm))
        # point2D index of observation (s,p) inside image s = rank among the masked points of that image
        masks = self.masks & self.registered[:, None]
        p2d_idx = np.cumsum(masks, axis=1) - 1
        reg = np.nonzero(self.registered)[0]
        with open(os.path.join(path, "images.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(reg)))
            for s in reg:
                q = rotmat_to_qvec(self.extrinsics[s, :, :3])
                t = self.extrinsics[s, :, 3]
                f.write(struct.pack("<i4d3di", s, *q, *t, 0 if self.shared_camera else s))
                f.write(self.image_names[s].encode() + b"\x00")
                pids = np.nonzero(masks[s])[0]
                f.write(struct.pack("<Q", len(pids)))
                rec = np.empty(len(pids), dtype=[("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
                rec["x"], rec["y"], rec["id"] = self.tracks[s, pids, 0], self.tracks[s, pids, 1], pids + 1
                f.write(rec.tobytes())
        with open(os.path.join(path, "points3D.bin"), "wb") as f:
            f.write(struct.pack("<Q", P + len(self.extra_points_xyz)))
            for p in range(P):
                frames = np.nonzero(masks[:, p])[0]
                f.write(struct.pack("<Q3d3Bd", p + 1, *self.points3D_xyz[p], *self.colors[p].tolist(), 0.0))
                f.write(struct.pack("<Q", len(frames)))
                tr = np.empty(len(frames), dtype=[("im", "<i4"), ("pt", "<i4")])
                tr["im"], tr["pt"] = frames, p2d_idx[frames, p]
                f.write(tr.tobytes())
            for k in range(len(self.extra_points_xyz)):
                f.write(struct.pack("<Q3d3Bd", P + 1 + k, *self.extra_points_xyz[k], *self.extra_points_rgb[k].tolist(), 0.0))
                f.write(struct.pack("<Q", 0))
