"""The `pycolmap` surface the reference touches, over flat arrays and the MI355X solvers (cut line B2).

Two uses:

* **the object the drop-in returns** -- ``vggsfm_amd.models.Triangulator`` and
  ``vggsfm_amd.utils.tensor_to_pycolmap.batch_matrix_to_pycolmap`` hand back a :class:`Reconstruction` built in bulk
  (:meth:`Reconstruction.from_arrays`, S numpy operations, no per-observation Python loop) that the reference's
  UNMODIFIED runner code can keep using: ``reconstruction.add_point3D(xyz, pycolmap.Track(), rgb)``
  (vggsfm/runners/runner.py:555-559), ``deregister_image`` (:575), ``images[id].name / .camera_id / .points2D[].xy``,
  ``cameras[id].params / .width / .height / .calibration_matrix()`` (:592-611, :1020-1052),
  ``points3D[id].track.elements`` (:758-762), ``write(dir)`` (:911);
* **a stand-in module** -- ``install()`` registers this module as ``sys.modules["pycolmap"]`` (and
  ``vggsfm_amd.pyceres_compat`` as ``pyceres``) so that the reference's own ``Triangulator.forward``,
  ``vggsfm/utils/triangulation.py``, ``tensor_to_pycolmap.py`` and ``VideoRunner`` run unmodified with
  ``bundle_adjustment`` / ``pose_refinement`` / ``absolute_pose_estimation`` / ``BundleAdjuster`` +
  ``pyceres.solve`` / ``ObservationManager`` executing on the GPU (``vgg_ba_solve``, ``vgg_pose_refine``,
  ``vgg_p3p_ransac``, ``vgg_filter_points``).  The reference's O(S*P) construction loops stay what they are.

Storage: a reconstruction keeps ONE copy of the observations -- per image a (m,2) float64 array of pixel
coordinates and a (m,) int64 array of point3D ids (:class:`ListPoint2D`) -- plus flat point arrays (xyz, colour,
error, alive).  ``points3D[id]``, ``images[id].points2D[k]`` and ``points3D[id].track`` are views into them; tracks
are derived from the per-image arrays (a cached CSR), which is exactly the invariant COLMAP maintains between
``Point2D.point3D_id`` and ``Track.elements``.

There is no CPU solver behind this module: the solver entries move the problem to ``DEVICE`` ("cuda") and fail
loudly without a GPU / without libvggsfm_amd.so.
"""
import os
import struct
import sys
import time
import weakref
from collections import namedtuple

import numpy as np
import torch

from .ba_options import (AbsolutePoseEstimationOptions, AbsolutePoseRefinementOptions,  # noqa: F401  (re-exported)
                         BundleAdjustmentOptions, RANSACOptions, SolverOptions, TERMINATION)

DEVICE = "cuda"
INVALID_POINT3D_ID = 2 ** 64 - 1            # pycolmap.INVALID_POINT3D_ID (kInvalidPoint3DId)
CAMERA_MODEL_IDS = {"SIMPLE_PINHOLE": 0, "SIMPLE_RADIAL": 2}
CAMERA_MODEL_NAMES = {v: k for k, v in CAMERA_MODEL_IDS.items()}
CAMERA_NUM_PARAMS = {"SIMPLE_PINHOLE": 3, "SIMPLE_RADIAL": 4}
__version__ = "3.10.0+vggsfm_amd"


def _np(x, dtype=np.float64):
    if torch.is_tensor(x):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=dtype)


# ----------------------------------------------------------------------------------------------- rigid transforms
def _quat_to_rotmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rotmat_to_qvec(R):
    """COLMAP qvec (w,x,y,z), w >= 0, of a rotation matrix (the order images.bin stores)."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(np.asarray(R, np.float64)).as_quat()        # x,y,z,w
    q = q[[3, 0, 1, 2]]
    return q if q[0] >= 0 else -q


class Rotation3d:
    """pycolmap.Rotation3d: built from a 3x3 matrix or a quaternion (x,y,z,w)."""

    def __init__(self, arg=None):
        if arg is None:
            self._R = np.eye(3)
            return
        a = _np(arg)
        if a.shape == (3, 3):
            self._R = a.copy()
        elif a.shape == (4,):
            self._R = _quat_to_rotmat(a / np.linalg.norm(a))
        else:
            raise ValueError("Rotation3d expects a 3x3 matrix or a quaternion (x,y,z,w)")

    def matrix(self):
        return self._R.copy()

    @property
    def quat(self):
        q = rotmat_to_qvec(self._R)
        return q[[1, 2, 3, 0]]

    def inverse(self):
        return Rotation3d(self._R.T)

    def __mul__(self, other):
        if isinstance(other, Rotation3d):
            return Rotation3d(self._R @ other._R)
        return self._R @ _np(other)


class Rigid3d:
    """pycolmap.Rigid3d(rotation, translation): x_cam = R x_world + t."""

    def __init__(self, rotation=None, translation=None):
        if rotation is not None and not isinstance(rotation, Rotation3d):
            m = _np(rotation)
            if m.shape == (3, 4) and translation is None:
                rotation, translation = Rotation3d(m[:, :3]), m[:, 3]
            else:
                rotation = Rotation3d(m)
        self.rotation = rotation if rotation is not None else Rotation3d()
        self.translation = np.zeros(3) if translation is None else _np(translation).reshape(3).copy()

    def matrix(self):
        return np.concatenate([self.rotation._R, self.translation[:, None]], axis=1)

    def inverse(self):
        Rt = self.rotation._R.T
        return Rigid3d(Rotation3d(Rt), -Rt @ self.translation)

    def __mul__(self, other):
        if isinstance(other, Rigid3d):
            return Rigid3d(Rotation3d(self.rotation._R @ other.rotation._R),
                           self.rotation._R @ other.translation + self.translation)
        return self.rotation._R @ _np(other) + self.translation


# ----------------------------------------------------------------------------------------------- camera
class Camera:
    """pycolmap.Camera for the two models the reference supports (SIMPLE_PINHOLE f,cx,cy; SIMPLE_RADIAL f,cx,cy,k)."""

    def __init__(self, model="SIMPLE_PINHOLE", width=0, height=0, params=(), camera_id=0):
        model = CAMERA_MODEL_NAMES.get(model, model) if isinstance(model, int) else str(getattr(model, "name", model))
        if model not in CAMERA_MODEL_IDS:
            raise ValueError(f"camera model {model} is not supported (SIMPLE_PINHOLE, SIMPLE_RADIAL)")
        self.model = model
        self.width, self.height = width, height
        self.params = params
        self.camera_id = int(camera_id)
        if len(self._params) != CAMERA_NUM_PARAMS[model]:
            raise ValueError(f"{model} takes {CAMERA_NUM_PARAMS[model]} parameters, got {len(self._params)}")

    model_name = property(lambda self: self.model)
    model_id = property(lambda self: CAMERA_MODEL_IDS[self.model])

    @property
    def params(self):
        return self._params

    @params.setter
    def params(self, v):
        self._params = np.array([float(p) for p in v], dtype=np.float64)

    @property
    def width(self):
        return self._w

    @width.setter
    def width(self, v):          # the reference assigns 0-d torch tensors (runner.py:1036-1037)
        self._w = int(v)

    @property
    def height(self):
        return self._h

    @height.setter
    def height(self, v):
        self._h = int(v)

    focal_length = property(lambda self: float(self._params[0]))
    principal_point_x = property(lambda self: float(self._params[1]))
    principal_point_y = property(lambda self: float(self._params[2]))

    def mean_focal_length(self):
        return float(self._params[0])

    def calibration_matrix(self):
        f, cx, cy = self._params[0], self._params[1], self._params[2]
        return np.array([[f, 0.0, cx], [0.0, f, cy], [0.0, 0.0, 1.0]])

    def _k(self):
        return self._params[3] if self.model == "SIMPLE_RADIAL" else 0.0

    def img_from_cam(self, cam_point):
        """Project: accepts a normalised 2-vector, a 3-vector in the camera frame (divided by z), or (N,2|3) arrays."""
        p = _np(cam_point)
        uv = p[..., :2] / p[..., 2:3] if p.shape[-1] == 3 else p
        r2 = (uv * uv).sum(-1, keepdims=True)
        return self._params[0] * uv * (1.0 + self._k() * r2) + self._params[1:3]

    def cam_from_img(self, image_point):
        """Unproject to the normalised plane (COLMAP's iterative undistortion for SIMPLE_RADIAL: Newton with a
        central-difference Jacobian, <= 100 iterations, stop at |step|^2 < 1e-10)."""
        x = (_np(image_point) - self._params[1:3]) / self._params[0]
        k = self._k()
        if k == 0.0:
            return x
        x0 = x.reshape(-1, 2)
        u = x0.copy()

        def dist(a):
            return a * (1.0 + k * (a * a).sum(-1, keepdims=True))

        for _ in range(100):
            step = np.maximum(np.finfo(np.float64).eps, np.abs(1e-6 * u))
            J = np.zeros(u.shape + (2,))
            for c in range(2):
                d = np.zeros_like(u)
                d[:, c] = step[:, c]
                J[:, :, c] = (dist(u + d) - dist(u - d)) / (2 * step[:, c:c + 1])
            dx = np.linalg.solve(J, (dist(u) - x0)[..., None])[..., 0]
            u = u - dx
            if (dx * dx).sum(-1).max() < 1e-10:
                break
        return u.reshape(x.shape)

    def __repr__(self):
        return f"Camera(camera_id={self.camera_id}, model={self.model}, width={self.width}, height={self.height}, params={self._params.tolist()})"


# ----------------------------------------------------------------------------------------------- 2D points
class Point2D:
    """pycolmap.Point2D(xy, point3D_id): a view into a :class:`ListPoint2D` (or its own one-row storage)."""
    __slots__ = ("_list", "_i")

    def __init__(self, xy=None, point3D_id=INVALID_POINT3D_ID):
        lst = ListPoint2D()
        lst._xy = np.zeros((1, 2)) if xy is None else _np(xy).reshape(1, 2).copy()
        pid = int(point3D_id)
        lst._pid = np.array([-1 if pid == INVALID_POINT3D_ID or pid < 0 else pid], dtype=np.int64)
        self._list, self._i = lst, 0

    @classmethod
    def _view(cls, lst, i):
        p = cls.__new__(cls)
        p._list, p._i = lst, i
        return p

    @property
    def xy(self):
        return self._list._xy[self._i].copy()

    @xy.setter
    def xy(self, v):
        self._list._xy[self._i] = _np(v).reshape(2)

    @property
    def point3D_id(self):
        pid = int(self._list._pid[self._i])
        return INVALID_POINT3D_ID if pid < 0 else pid

    @point3D_id.setter
    def point3D_id(self, v):
        v = int(v)
        self._list._pid[self._i] = -1 if v == INVALID_POINT3D_ID or v < 0 else v
        self._list._changed()

    def has_point3D(self):
        return bool(self._list._pid[self._i] >= 0)

    def __repr__(self):
        return f"Point2D(xy={self.xy.tolist()}, point3D_id={self.point3D_id})"


class ListPoint2D:
    """pycolmap.ListPoint2D: the 2D points of one image as two arrays -- xy (m,2) float64 and point3D ids (m,) int64
    (-1 = no 3D point).  Built from a list of :class:`Point2D` (the reference, tensor_to_pycolmap.py:147) or in bulk
    with :meth:`from_arrays`."""

    def __init__(self, points=()):
        self._owner = None
        if isinstance(points, ListPoint2D):
            self._xy, self._pid = points._xy.copy(), points._pid.copy()
            return
        points = list(points)
        if points:
            self._xy = np.concatenate([p._list._xy[p._i:p._i + 1] for p in points], axis=0)
            self._pid = np.array([p._list._pid[p._i] for p in points], dtype=np.int64)
        else:
            self._xy, self._pid = np.zeros((0, 2)), np.zeros(0, dtype=np.int64)

    @classmethod
    def from_arrays(cls, xy, point3D_ids):
        lst = cls()
        lst._xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        lst._pid = np.ascontiguousarray(point3D_ids, dtype=np.int64).reshape(-1)
        assert len(lst._xy) == len(lst._pid)
        return lst

    def _changed(self):
        img = self._owner() if self._owner is not None else None
        rec = img._rec() if (img is not None and img._rec is not None) else None
        if rec is not None:
            rec._tracks = None

    def __len__(self):
        return len(self._pid)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [Point2D._view(self, k) for k in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("ListPoint2D index out of range")
        return Point2D._view(self, i)

    def __iter__(self):
        return (Point2D._view(self, k) for k in range(len(self)))

    def append(self, p):
        self._xy = np.concatenate([self._xy, p._list._xy[p._i:p._i + 1]], axis=0)
        self._pid = np.concatenate([self._pid, p._list._pid[p._i:p._i + 1]])
        self._changed()

    def __repr__(self):
        return f"ListPoint2D(len={len(self)})"


class Image:
    """pycolmap.Image(id, name, camera_id, cam_from_world)."""

    def __init__(self, id=0, name="", camera_id=0, cam_from_world=None, image_id=None):
        self.image_id = int(id if image_id is None else image_id)
        self.name = name
        self.camera_id = int(camera_id)
        self.cam_from_world = cam_from_world if cam_from_world is not None else Rigid3d()
        self._rec = None
        self._registered = False
        self.points2D = ListPoint2D()

    @property
    def points2D(self):
        return self._points2D

    @points2D.setter
    def points2D(self, v):
        self._points2D = v if isinstance(v, ListPoint2D) else ListPoint2D(v)
        self._points2D._owner = weakref.ref(self)
        self._points2D._changed()

    @property
    def registered(self):
        return self._registered

    @registered.setter
    def registered(self, v):
        rec = self._rec() if self._rec is not None else None
        if rec is None:
            self._registered = bool(v)
        elif v:
            rec.register_image(self.image_id)
        else:
            rec.deregister_image(self.image_id)

    def num_points2D(self):
        return len(self._points2D)

    def num_points3D(self):
        return int((self._points2D._pid >= 0).sum())

    def projection_center(self):
        return -self.cam_from_world.rotation._R.T @ self.cam_from_world.translation

    def __repr__(self):
        return f"Image(image_id={self.image_id}, name={self.name!r}, camera_id={self.camera_id}, num_points2D={len(self._points2D)})"


# ----------------------------------------------------------------------------------------------- 3D points / tracks
TrackElement = namedtuple("TrackElement", ["image_id", "point2D_idx"])


class Track:
    """pycolmap.Track.  Free-standing (``pycolmap.Track()``) it owns a list of elements; obtained from
    ``reconstruction.points3D[id].track`` it is a view: ``elements`` is derived from the images' 2D points and
    ``add_element`` records the link on the image's 2D point (COLMAP keeps both sides in sync the same way)."""

    def __init__(self, elements=()):
        self._own = [TrackElement(int(e[0]), int(e[1])) for e in elements]
        self._rec, self._pid = None, None

    @classmethod
    def _view(cls, rec, pid):
        t = cls.__new__(cls)
        t._own, t._rec, t._pid = None, rec, pid
        return t

    @property
    def elements(self):
        if self._rec is None:
            return list(self._own)
        img, idx = self._rec._track_of(self._pid)
        return [TrackElement(int(a), int(b)) for a, b in zip(img, idx)]

    def add_element(self, image_id, point2D_idx):
        if self._rec is None:
            self._own.append(TrackElement(int(image_id), int(point2D_idx)))
        else:
            self._rec._link(int(image_id), int(point2D_idx), self._pid)

    def add_elements(self, elements):
        for e in elements:
            self.add_element(e[0], e[1])

    def length(self):
        return len(self._own) if self._rec is None else len(self._rec._track_of(self._pid)[0])

    def __len__(self):
        return self.length()


class Point3D:
    """View of one row of the reconstruction's point arrays (pycolmap.Point3D: xyz, color, error, track)."""
    __slots__ = ("_rec", "_pid")

    def __init__(self, rec, pid):
        self._rec, self._pid = rec, pid

    @property
    def xyz(self):
        # a COPY, as pycolmap returns (`p = rec.points3D[i].xyz; p *= s` must not edit the model; a view would also be
        # orphaned by the next _reserve()); writes go through the setter
        return self._rec._xyz[self._pid - 1].copy()

    @xyz.setter
    def xyz(self, v):
        self._rec._xyz[self._pid - 1] = _np(v).reshape(3)

    @property
    def color(self):
        return self._rec._rgb[self._pid - 1].copy()

    @color.setter
    def color(self, v):
        self._rec._rgb[self._pid - 1] = np.asarray(v).reshape(3).astype(np.uint8)

    @property
    def error(self):
        return float(self._rec._err[self._pid - 1])

    @error.setter
    def error(self, v):
        self._rec._err[self._pid - 1] = float(v)

    @property
    def track(self):
        return Track._view(self._rec, self._pid)

    def __repr__(self):
        return f"Point3D(xyz={self.xyz.tolist()}, track_length={self.track.length()})"


class _Point3DMap:
    """``reconstruction.points3D``: mapping point3D id (1-based) -> :class:`Point3D` over the live rows."""

    def __init__(self, rec):
        self._rec = rec

    def __len__(self):
        return int(self._rec._alive[:self._rec._n].sum())

    def __iter__(self):
        return iter((np.nonzero(self._rec._alive[:self._rec._n])[0] + 1).tolist())

    def __contains__(self, pid):
        pid = int(pid)
        return 1 <= pid <= self._rec._n and bool(self._rec._alive[pid - 1])

    def __getitem__(self, pid):
        pid = int(pid)
        if pid not in self:
            raise KeyError(pid)
        return Point3D(self._rec, pid)

    def __delitem__(self, pid):
        self._rec.delete_point3D(pid)

    def keys(self):
        return list(iter(self))

    def values(self):
        return [Point3D(self._rec, p) for p in self]

    def items(self):
        return [(p, Point3D(self._rec, p)) for p in self]


# ----------------------------------------------------------------------------------------------- reconstruction
class Reconstruction:
    def __init__(self, path=None):
        self.cameras, self.images = {}, {}
        self._reg = []                                     # registered image ids in registration order
        self._n = 0                                        # point3D ids are 1 .. _n (dead rows keep their slot)
        self._xyz = np.zeros((16, 3))
        self._rgb = np.zeros((16, 3), np.uint8)
        self._err = np.full(16, -1.0)
        self._alive = np.zeros(16, bool)
        self._pending = {}                                 # (image_id, point2D_idx) -> point3D id, image not added yet
        self._tracks = None                                # cached CSR (ptr, image ids, point2D indices)
        self.points3D = _Point3DMap(self)
        if path is not None:
            self.read(path)

    # ---- bulk construction (the drop-in's batch_matrix_to_pycolmap)
    @classmethod
    def from_arrays(cls, points3d, extrinsics, intrinsics, tracks, masks, image_size, max_points3D_val=3000,
                    shared_camera=False, camera_type="SIMPLE_PINHOLE", extra_params=None, colors=None):
        """Same selection rules as the reference's loop (tensor_to_pycolmap.py:62-158): a track enters when it has >= 2
        masked observations (ids 1.. in track order); observations of points with a coordinate >= max_points3D_val are
        skipped; one camera per frame or a single one carrying frame 0's parameters; every image registered.
        numpy inputs: points3d (P,3), extrinsics (S,3,4), intrinsics (S,3,3), tracks (S,P,2), masks (S,P)."""
        if camera_type not in CAMERA_MODEL_IDS:
            raise ValueError(f"Camera type {camera_type} is not supported yet")
        pts, ext, K = _np(points3d), _np(extrinsics), _np(intrinsics)
        trk, msk = _np(tracks, None), _np(masks, None).astype(bool)
        size = _np(image_size).reshape(-1)
        S = len(ext)
        rec = cls()
        valid_idx = np.nonzero(msk.sum(0) >= 2)[0]
        rec.valid_idx = valid_idx
        n = len(valid_idx)
        rec._reserve(n)
        rec._xyz[:n], rec._alive[:n], rec._n = pts[valid_idx], True, n
        if colors is not None:
            rec._rgb[:n] = np.asarray(colors).reshape(-1, 3)[:n].astype(np.uint8)
        ok = (rec._xyz[:n] < max_points3D_val).all(-1)
        camera = None
        for f in range(S):
            if camera is None or not shared_camera:
                prm = [K[f, 0, 0], K[f, 0, 2], K[f, 1, 2]]
                if camera_type == "SIMPLE_RADIAL":
                    prm.append(_np(extra_params)[f, 0])
                camera = Camera(camera_type, size[0], size[1], prm, f)
                rec.add_camera(camera)
            sel = np.nonzero(msk[f, valid_idx] & ok)[0]
            img = Image(f, f"image_{f}", camera.camera_id, Rigid3d(Rotation3d(ext[f, :, :3]), ext[f, :, 3]))
            img.points2D = ListPoint2D.from_arrays(trk[f, valid_idx[sel]], sel + 1)
            img._registered = True
            rec.add_image(img)
        return rec

    @classmethod
    def from_frame_lists(cls, points3d_valid, valid_idx, extrinsics, intrinsics, xy, point3D_ids, frame_counts, image_size,
                         shared_camera=False, camera_type="SIMPLE_PINHOLE", extra_params=None):
        """The same model as :meth:`from_arrays` from observation LISTS the caller selected already (on the device:
        ``vggsfm_amd.utils.tensor_to_pycolmap.batch_matrix_to_pycolmap``): points3d_valid (n,3) = the kept tracks in track
        order, valid_idx (n,) their input track indices, xy (O,2) f64 / point3D_ids (O,) i64 frame-major (frame f owns the
        next frame_counts[f] rows, tracks ascending inside a frame).  The per-image lists are views of xy / point3D_ids."""
        if camera_type not in CAMERA_MODEL_IDS:
            raise ValueError(f"Camera type {camera_type} is not supported yet")
        ext, K, size = _np(extrinsics), _np(intrinsics), _np(image_size).reshape(-1)
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        pid = np.ascontiguousarray(point3D_ids, dtype=np.int64).reshape(-1)
        S = len(ext)
        rec = cls()
        rec.valid_idx = np.asarray(valid_idx)
        n = len(points3d_valid)
        rec._reserve(n)
        rec._xyz[:n], rec._alive[:n], rec._n = _np(points3d_valid), True, n
        off = np.concatenate([[0], np.cumsum(np.asarray(frame_counts, dtype=np.int64))])
        assert off[-1] == len(pid) == len(xy) and len(off) == S + 1
        camera = None
        for f in range(S):
            if camera is None or not shared_camera:
                prm = [K[f, 0, 0], K[f, 0, 2], K[f, 1, 2]]
                if camera_type == "SIMPLE_RADIAL":
                    prm.append(_np(extra_params)[f, 0])
                camera = Camera(camera_type, size[0], size[1], prm, f)
                rec.add_camera(camera)
            img = Image(f, f"image_{f}", camera.camera_id, Rigid3d(Rotation3d(ext[f, :, :3]), ext[f, :, 3]))
            img.points2D = ListPoint2D.from_arrays(xy[off[f]:off[f + 1]], pid[off[f]:off[f + 1]])
            img._registered = True
            rec.add_image(img)
        return rec

    # ---- point storage
    def _reserve(self, n):
        cap = len(self._alive)
        if n <= cap:
            return
        new = max(n, 2 * cap)
        for name, fill in (("_xyz", 0.0), ("_rgb", 0), ("_err", -1.0), ("_alive", False)):
            old = getattr(self, name)
            arr = np.full((new,) + old.shape[1:], fill, dtype=old.dtype)
            arr[:cap] = old
            setattr(self, name, arr)

    def add_point3D(self, xyz, track=None, color=np.zeros(3)):
        """``reconstruction.add_point3D(xyz, pycolmap.Track(), color)`` -> the new id (ids are consecutive from 1)."""
        self._reserve(self._n + 1)
        i = self._n
        self._xyz[i] = _np(xyz).reshape(3)
        self._rgb[i] = np.asarray(color).reshape(3).astype(np.uint8)
        self._err[i], self._alive[i] = -1.0, True
        self._n += 1
        self._tracks = None
        if track is not None:
            for e in track.elements:
                self._link(e.image_id, e.point2D_idx, self._n)
        return self._n

    def add_points3D(self, xyz, colors=None):
        """Bulk form of ``add_point3D(xyz, Track(), color)`` (points without observations) -> first new id."""
        xyz = _np(xyz).reshape(-1, 3)
        k = len(xyz)
        self._reserve(self._n + k)
        s = slice(self._n, self._n + k)
        self._xyz[s], self._err[s], self._alive[s] = xyz, -1.0, True
        self._rgb[s] = 0 if colors is None else np.asarray(colors).reshape(-1, 3).astype(np.uint8)
        self._n += k
        self._tracks = None
        return self._n - k + 1

    def set_colors(self, colors):
        """Colours of points 1..len(colors) at once (the loop of vggsfm/models/triangulator.py:335-342)."""
        c = np.asarray(colors).reshape(-1, 3).astype(np.uint8)
        self._rgb[:len(c)] = c

    def delete_point3D(self, point3D_id):
        self._delete_points(np.array([int(point3D_id)], dtype=np.int64))

    def _delete_points(self, pids):
        pids = np.asarray(pids, dtype=np.int64)
        pids = pids[self._alive[pids - 1]] if len(pids) else pids
        if len(pids) == 0:
            return
        dead = np.zeros(self._n + 1, bool)
        dead[pids] = True
        for im in self.images.values():                   # reset the 2D points that referenced them
            pid = im.points2D._pid
            hit = (pid >= 0) & dead[np.clip(pid, 0, self._n)]
            if hit.any():
                pid[hit] = -1
        self._alive[pids - 1] = False
        self._tracks = None

    def _delete_observations(self, image_ids, point2D_idxs):
        """``DeleteObservation`` for many (image, 2D point) pairs, evaluated against the tracks as they are NOW: a point
        whose track would fall below two elements goes as a whole (COLMAP deletes it when an element is removed from a
        track of length <= 2)."""
        image_ids, point2D_idxs = np.asarray(image_ids, np.int64), np.asarray(point2D_idxs, np.int64)
        if len(image_ids) == 0:
            return
        ptr, _, _ = self._track_csr()
        length = np.diff(ptr)                              # per point id - 1
        pids = np.array([self.images[int(i)].points2D._pid[int(k)] for i, k in zip(image_ids, point2D_idxs)], np.int64) \
            if len(image_ids) < 64 else self._pids_of(image_ids, point2D_idxs)
        keep = pids >= 0
        image_ids, point2D_idxs, pids = image_ids[keep], point2D_idxs[keep], pids[keep]
        nrem = np.bincount(pids, minlength=self._n + 1)
        dead = np.nonzero((nrem[1:] > 0) & (length - nrem[1:] < 2))[0] + 1
        for i in np.unique(image_ids):
            sel = image_ids == i
            self.images[int(i)].points2D._pid[point2D_idxs[sel]] = -1
        self._tracks = None
        self._delete_points(dead)

    def _pids_of(self, image_ids, point2D_idxs):
        out = np.empty(len(image_ids), np.int64)
        for i in np.unique(image_ids):
            sel = image_ids == i
            out[sel] = self.images[int(i)].points2D._pid[point2D_idxs[sel]]
        return out

    # ---- tracks derived from the images' 2D points
    def _track_csr(self):
        if self._tracks is None:
            pid_l, img_l, idx_l = [], [], []
            for i in sorted(self.images):
                pid = self.images[i].points2D._pid
                k = np.nonzero(pid >= 0)[0]
                k = k[self._alive[np.clip(pid[k] - 1, 0, len(self._alive) - 1)] & (pid[k] <= self._n)]
                pid_l.append(pid[k])
                img_l.append(np.full(len(k), i, np.int64))
                idx_l.append(k)
            pid = np.concatenate(pid_l) if pid_l else np.zeros(0, np.int64)
            img = np.concatenate(img_l) if img_l else np.zeros(0, np.int64)
            idx = np.concatenate(idx_l) if idx_l else np.zeros(0, np.int64)
            order = np.argsort(pid, kind="stable")
            ptr = np.zeros(self._n + 1, np.int64)
            np.cumsum(np.bincount(pid, minlength=self._n + 1)[1:], out=ptr[1:])
            self._tracks = (ptr, img[order], idx[order])
        return self._tracks

    def _track_of(self, pid):
        ptr, img, idx = self._track_csr()
        a, b = ptr[pid - 1], ptr[pid]
        return img[a:b], idx[a:b]

    def _link(self, image_id, point2D_idx, pid):
        im = self.images.get(image_id)
        if im is None or point2D_idx >= len(im.points2D):
            self._pending[(image_id, point2D_idx)] = pid    # applied when the image (or its 2D points) arrives
            return
        if im.points2D._pid[point2D_idx] != pid:
            im.points2D._pid[point2D_idx] = pid
            self._tracks = None

    # ---- cameras / images
    def add_camera(self, camera):
        self.cameras[camera.camera_id] = camera

    def add_image(self, image):
        self.images[image.image_id] = image
        image._rec = weakref.ref(self)
        if self._pending:
            for (i, k), pid in [kv for kv in self._pending.items() if kv[0][0] == image.image_id]:
                if k < len(image.points2D) and image.points2D._pid[k] < 0:
                    image.points2D._pid[k] = pid
                del self._pending[(i, k)]
        if image._registered and image.image_id not in self._reg:
            self._reg.append(image.image_id)
        self._tracks = None

    def register_image(self, image_id):
        image_id = int(image_id)
        self.images[image_id]._registered = True
        if image_id not in self._reg:
            self._reg.append(image_id)

    def deregister_image(self, image_id):
        """COLMAP ``Reconstruction::DeRegisterImage``: every observation of the image is deleted (points left with a
        single observation go with it), the image stays in the model without being registered."""
        image_id = int(image_id)
        im = self.images[image_id]
        k = np.nonzero(im.points2D._pid >= 0)[0]
        self._delete_observations(np.full(len(k), image_id, np.int64), k)
        im._registered = False
        if image_id in self._reg:
            self._reg.remove(image_id)

    def reg_image_ids(self):
        return list(self._reg)

    def point3D_ids(self):
        return set(iter(self.points3D))

    def num_points3D(self):
        return len(self.points3D)

    def num_images(self):
        return len(self.images)

    def num_reg_images(self):
        return len(self._reg)

    def num_cameras(self):
        return len(self.cameras)

    def exists_point3D(self, pid):
        return pid in self.points3D

    def is_image_registered(self, image_id):
        return int(image_id) in self._reg

    def summary(self):
        obs = sum(self.images[i].num_points3D() for i in self._reg)
        return (f"Reconstruction:\n\tnum_reg_images = {len(self._reg)}\n\tnum_cameras = {len(self.cameras)}\n"
                f"\tnum_points3D = {len(self.points3D)}\n\tnum_observations = {obs}")

    def __repr__(self):
        return f"Reconstruction(num_reg_images={len(self._reg)}, num_cameras={len(self.cameras)}, num_points3D={len(self.points3D)})"

    def __deepcopy__(self, memo=None):
        new = Reconstruction()
        new._n = self._n
        for name in ("_xyz", "_rgb", "_err", "_alive"):
            setattr(new, name, getattr(self, name).copy())
        for cid, c in self.cameras.items():
            new.add_camera(Camera(c.model, c.width, c.height, c.params, cid))
        for i, im in self.images.items():
            cp = Image(i, im.name, im.camera_id, Rigid3d(Rotation3d(im.cam_from_world.rotation._R), im.cam_from_world.translation))
            cp.points2D = ListPoint2D(im.points2D)
            cp._registered = im._registered
            new.add_image(cp)
        new._reg = list(self._reg)
        new._pending = dict(self._pending)
        if hasattr(self, "valid_idx"):
            new.valid_idx = self.valid_idx
        return new

    # ---- similarity normalisation
    def normalize(self, extent=10.0, p0=0.1, p1=0.9, use_images=True):
        """``Reconstruction::Normalize`` (COLMAP 3.10) with use_images: percentile box of the registered camera centres
        (sorted as float32), scale to `extent`, centre on the mean; applied to the registered poses and every point."""
        if not use_images:
            raise NotImplementedError("normalize(use_images=False) is not used by the reference")
        from .ba import normalization_transform
        ids = list(self._reg)
        if len(ids) < 2:
            return
        ext = torch.from_numpy(np.stack([self.images[i].cam_from_world.matrix() for i in ids]))
        scale, mean, ext2 = normalization_transform(ext, extent, p0, p1)
        ext2 = ext2.numpy()
        # the points on the host arrays they live in (same two roundings per coordinate as the tensor version)
        alive = self._alive[:self._n]
        self._xyz[:self._n][alive] = float(scale) * (self._xyz[:self._n][alive] - mean.numpy())
        for k, i in enumerate(ids):
            self.images[i].cam_from_world = Rigid3d(Rotation3d(ext2[k, :, :3]), ext2[k, :, 3])

    # ---- the flat problem the solvers take
    def problem_arrays(self, image_ids=None):
        """(image_ids, points (n,3), alive (n,), extrinsics (S,3,4), K (S,3,3), extra (S,1)|None, tracks (S,n,2) f64,
        masks (S,n) bool, shared_camera, camera_type) over the given (default: registered) images."""
        ids = list(self._reg) if image_ids is None else [int(i) for i in image_ids]
        S, n = len(ids), self._n
        tracks = np.zeros((S, n, 2))
        masks = np.zeros((S, n), bool)
        for r, i in enumerate(ids):
            p2 = self.images[i].points2D
            k = np.nonzero(p2._pid >= 0)[0]
            k = k[self._alive[p2._pid[k] - 1]]
            tracks[r, p2._pid[k] - 1] = p2._xy[k]
            masks[r, p2._pid[k] - 1] = True
        ext = np.stack([self.images[i].cam_from_world.matrix() for i in ids]) if S else np.zeros((0, 3, 4))
        cams = [self.cameras[self.images[i].camera_id] for i in ids]
        cam_ids = {c.camera_id for c in cams}
        shared = len(cam_ids) == 1 and S > 1
        if not shared and len(cam_ids) != S:
            raise NotImplementedError("cameras shared by some but not all images are not supported")
        models = {c.model for c in cams}
        if len(models) > 1:
            raise NotImplementedError("mixed camera models in one problem are not supported")
        model = cams[0].model if cams else "SIMPLE_PINHOLE"
        K = np.stack([c.calibration_matrix() for c in cams]) if S else np.zeros((0, 3, 3))
        extra = np.stack([c.params[3:4] for c in cams]) if model == "SIMPLE_RADIAL" else None
        return ids, self._xyz[:n].copy(), self._alive[:n].copy(), ext, K, extra, tracks, masks, shared, model

    def _store_cameras(self, ids, ext, K, extra):
        for r, i in enumerate(ids):
            im = self.images[i]
            im.cam_from_world = Rigid3d(Rotation3d(ext[r, :, :3]), ext[r, :, 3])
            cam = self.cameras[im.camera_id]
            cam._params[0], cam._params[1], cam._params[2] = K[r, 0, 0], K[r, 0, 2], K[r, 1, 2]
            if extra is not None:
                cam._params[3] = extra[r, 0]

    # ---- COLMAP binary model (format as read by vggsfm/datasets/imc_helper.py:127-159, 238-286, 380-416)
    def write(self, path):
        self.write_binary(path)

    def write_binary(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "cameras.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(self.cameras)))
            for cid in sorted(self.cameras):
                c = self.cameras[cid]
                f.write(struct.pack("<iiQQ", cid, CAMERA_MODEL_IDS[c.model], c.width, c.height))
                f.write(c.params.astype("<f8").tobytes())
        with open(os.path.join(path, "images.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(self._reg)))
            for i in sorted(self._reg):
                im = self.images[i]
                q = rotmat_to_qvec(im.cam_from_world.rotation._R)
                f.write(struct.pack("<i4d3di", i, *q, *im.cam_from_world.translation, im.camera_id))
                f.write(str(im.name).encode() + b"\x00")
                p2 = im.points2D
                rec = np.empty(len(p2), dtype=[("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
                rec["x"], rec["y"] = p2._xy[:, 0], p2._xy[:, 1]
                pid = p2._pid.copy()
                pid[(pid >= 0) & ~self._alive[np.clip(pid - 1, 0, len(self._alive) - 1)]] = -1
                rec["id"] = pid
                f.write(struct.pack("<Q", len(p2)))
                f.write(rec.tobytes())
        # points3D.bin: 43-byte header + 8 bytes per track element, assembled as one byte buffer
        ptr, timg, tidx = self._track_csr()
        ids = np.nonzero(self._alive[:self._n])[0]
        tlen = (ptr[ids + 1] - ptr[ids]).astype(np.int64)
        HEAD = 8 + 24 + 3 + 8 + 8
        off = np.zeros(len(ids) + 1, np.int64)
        np.cumsum(HEAD + 8 * tlen, out=off[1:])
        buf = np.zeros(int(off[-1]), np.uint8)

        def put(col, arr):
            b = np.ascontiguousarray(arr).view(np.uint8).reshape(len(ids), -1)
            buf[(off[:-1, None] + col + np.arange(b.shape[1])[None]).ravel()] = b.ravel()

        if len(ids):
            put(0, (ids + 1).astype("<u8"))
            put(8, self._xyz[ids].astype("<f8"))
            put(32, self._rgb[ids])
            put(35, np.where(self._err[ids] < 0, 0.0, self._err[ids]).astype("<f8"))
            put(43, tlen.astype("<u8"))
            tot = int(tlen.sum())
            if tot:
                src = (np.repeat(ptr[ids], tlen) + (np.arange(tot) - np.repeat(np.cumsum(tlen) - tlen, tlen)))
                el = np.empty(tot, dtype=[("im", "<i4"), ("pt", "<i4")])
                el["im"], el["pt"] = timg[src], tidx[src]
                dst = np.repeat(off[:-1] + HEAD, tlen) + 8 * (np.arange(tot) - np.repeat(np.cumsum(tlen) - tlen, tlen))
                buf[(dst[:, None] + np.arange(8)[None]).ravel()] = el.view(np.uint8).ravel()
        with open(os.path.join(path, "points3D.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(ids)))
            f.write(buf.tobytes())

    def read(self, path):
        self.read_binary(path)

    def read_binary(self, path):
        """Load cameras.bin / images.bin / points3D.bin written by COLMAP (or by :meth:`write`)."""
        with open(os.path.join(path, "cameras.bin"), "rb") as f:
            (n,) = struct.unpack("<Q", f.read(8))
            for _ in range(n):
                cid, mid, w, h = struct.unpack("<iiQQ", f.read(24))
                model = CAMERA_MODEL_NAMES[mid]
                k = CAMERA_NUM_PARAMS[model]
                self.add_camera(Camera(model, w, h, struct.unpack(f"<{k}d", f.read(8 * k)), cid))
        with open(os.path.join(path, "images.bin"), "rb") as f:
            (n,) = struct.unpack("<Q", f.read(8))
            images = []
            for _ in range(n):
                v = struct.unpack("<i4d3di", f.read(64))
                name = b""
                while True:
                    ch = f.read(1)
                    if ch == b"\x00":
                        break
                    name += ch
                (m,) = struct.unpack("<Q", f.read(8))
                rec = np.frombuffer(f.read(24 * m), dtype=[("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
                q = np.array(v[1:5])
                img = Image(v[0], name.decode(), v[8], Rigid3d(Rotation3d(q[[1, 2, 3, 0]]), np.array(v[5:8])))
                img.points2D = ListPoint2D.from_arrays(np.stack([rec["x"], rec["y"]], 1), rec["id"])
                img._registered = True
                images.append(img)
        with open(os.path.join(path, "points3D.bin"), "rb") as f:
            (n,) = struct.unpack("<Q", f.read(8))
            recs = []
            for _ in range(n):
                pid, x, y, z, r, g, b, err, ln = struct.unpack("<Q3d3BdQ", f.read(51))
                f.read(8 * ln)                              # track elements: implied by the images' 2D points
                recs.append((pid, (x, y, z), (r, g, b), err))
        top = max((r[0] for r in recs), default=0)
        self._reserve(top)
        self._n = top
        for pid, xyz, rgb, err in recs:
            self._xyz[pid - 1], self._rgb[pid - 1], self._err[pid - 1], self._alive[pid - 1] = xyz, rgb, err, True
        for img in images:
            self.add_image(img)


# ----------------------------------------------------------------------------------------------- options / config
class BundleAdjustmentConfig:
    """pycolmap.BundleAdjustmentConfig (video_runner.py:817-829)."""

    def __init__(self):
        self.image_ids, self.constant_cam_poses = [], set()
        self.constant_cam_positions = {}
        self.constant_point3D_ids, self.variable_point3D_ids = set(), set()

    def add_image(self, image_id):
        if int(image_id) not in self.image_ids:
            self.image_ids.append(int(image_id))

    def set_constant_cam_pose(self, image_id):
        self.constant_cam_poses.add(int(image_id))

    def set_constant_cam_positions(self, image_id, idxs):
        self.constant_cam_positions[int(image_id)] = list(idxs)

    def add_constant_point(self, point3D_id):
        self.constant_point3D_ids.add(int(point3D_id))

    def add_variable_point(self, point3D_id):
        self.variable_point3D_ids.add(int(point3D_id))

    def num_images(self):
        return len(self.image_ids)


def _create_loss_function(self):
    """``BundleAdjustmentOptions.create_loss_function()``: the solver reads type and scale from the options."""
    return (self.loss_function_type, self.loss_function_scale)


BundleAdjustmentOptions.create_loss_function = _create_loss_function


# ----------------------------------------------------------------------------------------------- solvers (device)
def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEVICE)
    return t if dtype is None else t.to(dtype)


def _filter_negative_depth(rec, ids=None):
    """``ObservationManager::FilterObservationsWithNegativeDepth``: delete the observations whose point has depth
    < eps in their image (whole points when fewer than two observations would remain).  Returns the number deleted."""
    img_l, idx_l = [], []
    for i in (rec.reg_image_ids() if ids is None else ids):
        im = rec.images[i]
        p2 = im.points2D
        k = np.nonzero(p2._pid >= 0)[0]
        k = k[rec._alive[p2._pid[k] - 1]]
        M = im.cam_from_world.matrix()
        z = rec._xyz[p2._pid[k] - 1] @ M[2, :3] + M[2, 3]
        bad = k[~(z >= np.finfo(np.float64).eps)]
        img_l.append(np.full(len(bad), i, np.int64))
        idx_l.append(bad)
    img = np.concatenate(img_l) if img_l else np.zeros(0, np.int64)
    rec._delete_observations(img, np.concatenate(idx_l) if idx_l else img)
    return len(img)


def _solve(rec, options, ids, constant_points=None, constant_pose_ids=None, default_gauge=True):
    from . import ba as _ba
    ids, pts, alive, ext, K, extra, tracks, masks, shared, model = rec.problem_arrays(ids)
    if constant_pose_ids is not None:
        constant_pose_frames = [ids.index(i) for i in constant_pose_ids if i in ids]
    cp = None
    if constant_points is not None:
        cp = np.zeros(len(pts), bool)
        c = np.array(sorted(p for p in constant_points if 1 <= p <= len(pts)), dtype=np.int64)
        cp[c - 1] = True
    t0 = time.time()
    p_opt, e_opt, K_opt, x_opt, summ = _ba.bundle_adjustment(
        _dev(pts), _dev(ext), _dev(K), _dev(tracks, torch.float32), _dev(masks), None,
        None if extra is None else _dev(extra), shared, model, options, False,
        constant_points=None if cp is None else _dev(cp),
        constant_pose_frames=None if default_gauge else constant_pose_frames, filter_negative_depth=False)
    vi = summ["valid_idx"].cpu().numpy()
    rec._xyz[vi] = p_opt.cpu().numpy()
    rec._store_cameras(ids, e_opt.cpu().numpy(), K_opt.cpu().numpy(), None if x_opt is None else x_opt.cpu().numpy())
    summ["total_time_in_seconds"] = time.time() - t0
    summ["num_residuals"] = 2 * int(masks[:, vi].sum())
    return summ


def bundle_adjustment(reconstruction, options=None):
    """``pycolmap.bundle_adjustment`` = COLMAP's BundleAdjustmentController: negative-depth observation filter, all
    registered images, first image's pose and the second image's t_x constant, Ceres LM -- on the GPU
    (``vggsfm_amd.ba.bundle_adjustment`` -> ``vgg_ba_solve``).  The reconstruction is updated in place."""
    options = options or BundleAdjustmentOptions()
    _filter_negative_depth(reconstruction)
    return _solve(reconstruction, options, None)


class _Problem:
    """What ``BundleAdjuster.problem`` hands to ``pyceres.solve``."""

    def __init__(self, adjuster):
        self.adjuster = adjuster

    def num_residuals(self):
        rec = self.adjuster.reconstruction
        return 2 * sum(rec.images[i].num_points3D() for i in self.adjuster.config.image_ids)


class BundleAdjuster:
    """pycolmap.BundleAdjuster(options, config): ``set_up_problem`` + ``pyceres.solve`` (video_runner.py:1321-1331)."""

    def __init__(self, options, config):
        self.options, self.config = options, config
        self.reconstruction, self.problem = None, None

    def set_up_problem(self, reconstruction, loss_function=None):
        if self.config.constant_cam_positions:           # (loud at set-up, not only when the solve runs)
            raise NotImplementedError("BundleAdjustmentConfig.set_constant_cam_positions is not used by the reference")
        self.reconstruction = reconstruction
        if loss_function is not None and isinstance(loss_function, tuple):
            self.options.loss_function_type, self.options.loss_function_scale = loss_function
        self.problem = _Problem(self)

    def set_up_solver_options(self, problem, solver_options):
        return solver_options

    def _run(self, solver_options):
        import copy
        opts = copy.copy(self.options)
        opts.solver_options = solver_options
        cfg = self.config
        if cfg.constant_cam_positions:
            raise NotImplementedError("BundleAdjustmentConfig.set_constant_cam_positions is not used by the reference")
        # COLMAP BundleAdjuster::SetUp: a point whose track has observations in images OUTSIDE the config is held constant
        # (its track length exceeds its observations inside the problem) unless it was named with add_variable_point.  The
        # reference's call sites put every registered image into the config (video_runner.py:817-829), where this selects
        # nothing.  COLMAP's tracks only hold REGISTERED images (DeRegisterImage deletes the observations): 2D points linked
        # in an image that is not registered do not count.  A variable point with observations outside the config would need
        # those images in the problem with constant poses (COLMAP adds them); the reference never builds such a config, so
        # that case raises instead of silently losing the observations.
        rec = self.reconstruction
        partly_outside = set()
        if not set(rec.reg_image_ids()) <= set(cfg.image_ids):    # (every reference call site: the config holds them all)
            ptr, timg, _ = rec._track_csr()
            registered = np.isin(timg, np.asarray(rec.reg_image_ids(), np.int64))
            inside = np.isin(timg, np.asarray(cfg.image_ids, np.int64))
            pid_of_obs = np.repeat(np.arange(1, len(ptr)), np.diff(ptr))
            partly_outside = {int(p) for p in np.unique(pid_of_obs[registered & ~inside])}
        lost = partly_outside & set(cfg.variable_point3D_ids)
        if lost:
            raise NotImplementedError(f"BundleAdjustmentConfig: {len(lost)} variable point(s) (e.g. id {min(lost)}) are observed in "
                                      "registered images outside the config; COLMAP would add those observations with constant "
                                      "poses -- add the images to the config with set_constant_cam_pose")
        constant = set(cfg.constant_point3D_ids) | partly_outside
        return _solve(rec, opts, cfg.image_ids, constant, sorted(cfg.constant_cam_poses), default_gauge=False)

    def solve(self, reconstruction):
        self.set_up_problem(reconstruction)
        return self._run(self.options.solver_options)


def _camera_params4(camera):
    p = np.zeros(4)
    p[:len(camera.params)] = camera.params
    return p


def pose_refinement(cam_from_world, points2D, points3D, inlier_mask, camera, refinement_options=None):
    """``pycolmap.pose_refinement`` (RefineAbsolutePose: Cauchy loss, points constant) for one image on the GPU
    (``vgg_pose_refine``) -> {"cam_from_world": Rigid3d, "num_inliers"}; `camera.params` are refined in place."""
    from .pose import pose_refinement_batch
    ro = refinement_options or AbsolutePoseRefinementOptions()
    flags = torch.tensor([int(bool(ro.refine_focal_length)) | (int(bool(ro.refine_extra_params)) << 1)],
                         dtype=torch.uint8, device=DEVICE)
    mask = np.asarray(inlier_mask, bool).reshape(1, -1)
    ext, intr, sums = pose_refinement_batch(_dev(cam_from_world.matrix()[None]), _dev(_camera_params4(camera)[None]),
                                            _dev(_np(points2D)[None]), _dev(_np(points3D)), _dev(mask), [0],
                                            camera.model, flags, ro)
    camera.params = intr[0, :len(camera.params)].cpu().numpy()
    e = ext[0].cpu().numpy()
    return {"cam_from_world": Rigid3d(Rotation3d(e[:, :3]), e[:, 3]), "num_inliers": int(mask.sum()), "summary": sums[0]}


def absolute_pose_estimation(points2D, points3D, camera, estimation_options=None, refinement_options=None,
                             return_covariance=False):
    """``pycolmap.absolute_pose_estimation``: P3P RANSAC (``vgg_p3p_ransac``; over COLMAP's focal length factors when
    `estimate_focal_length`) + refinement on the inliers.  -> {"cam_from_world", "num_inliers", "inliers"} or None."""
    from .pose import absolute_pose_estimation_batch
    eo = estimation_options or AbsolutePoseEstimationOptions()
    ro = refinement_options or AbsolutePoseRefinementOptions()
    p2, p3 = _np(points2D).reshape(-1, 2), _np(points3D).reshape(-1, 3)
    if len(p2) < 3:
        return None
    flags = torch.tensor([int(bool(ro.refine_focal_length)) | (int(bool(ro.refine_extra_params)) << 1)],
                         dtype=torch.uint8, device=DEVICE)
    ext, intr, ok, num, inl = absolute_pose_estimation_batch(
        torch.eye(3, 4, dtype=torch.float64, device=DEVICE)[None], _dev(_camera_params4(camera)[None]), _dev(p2[None]),
        _dev(p3), torch.ones((1, len(p2)), dtype=torch.bool, device=DEVICE), [0], camera.model, flags, eo, ro)
    if not bool(ok[0]):
        return None
    camera.params = intr[0, :len(camera.params)].cpu().numpy()
    e = ext[0].cpu().numpy()
    return {"cam_from_world": Rigid3d(Rotation3d(e[:, :3]), e[:, 3]), "num_inliers": int(num[0]),
            "inliers": inl[0].cpu().numpy()}


class ObservationManager:
    """pycolmap.ObservationManager(reconstruction): the two filters the reference calls (video_runner.py:510-512,
    triangulation.py:1214-1216), evaluated by ``vgg_filter_points`` on the GPU and applied to the arrays."""

    def __init__(self, reconstruction, correspondence_graph=None):
        self.reconstruction = reconstruction

    def filter_observations_with_negative_depth(self):
        return _filter_negative_depth(self.reconstruction)

    def filter_all_points3D(self, max_reproj_error, min_tri_angle):
        """FilterPoints3DWithLargeReprojectionError + FilterPoints3DWithSmallTriangulationAngle [COLMAP 3.10,
        observation_manager.cc]: an observation goes when its squared reprojection error exceeds max^2 or its depth is
        not positive; a point goes when fewer than two observations remain or when no pair of its remaining views
        subtends at least `min_tri_angle` degrees.  Returns the number of filtered observations."""
        from .video import observation_filter
        rec = self.reconstruction
        ids, pts, alive, ext, K, extra, tracks, masks, shared, model = rec.problem_arrays()
        inl, keep = observation_filter(_dev(pts), _dev(ext), _dev(K), None if extra is None else _dev(extra),
                                       _dev(tracks), _dev(masks), max_reproj_error, min_tri_angle)
        inl, keep = inl.cpu().numpy(), keep.cpu().numpy()
        before = int(masks.sum())
        img_l, idx_l = [], []
        for r, i in enumerate(ids):
            p2 = rec.images[i].points2D
            k = np.nonzero(p2._pid >= 0)[0]
            k = k[rec._alive[p2._pid[k] - 1]]
            bad = k[~inl[r, p2._pid[k] - 1]]
            p2._pid[bad] = -1
        rec._tracks = None
        rec._delete_points(np.nonzero(alive & ~keep)[0] + 1)
        return before - int((inl & keep[None]).sum())


# ----------------------------------------------------------------------------------------------- module registration
def install(force=False):
    """Register this module as ``pycolmap`` and ``vggsfm_amd.pyceres_compat`` as ``pyceres`` (only when the real
    packages are not importable, unless `force`), so that the reference's own geometry code runs on the GPU solvers."""
    from . import pyceres_compat
    for name, mod in (("pycolmap", sys.modules[__name__]), ("pyceres", pyceres_compat)):
        if not force:
            try:
                __import__(name)
                continue
            except ImportError:
                pass
        sys.modules[name] = mod
