"""A `pycolmap`-shaped module backed by the CPU oracle (TEST INFRASTRUCTURE ONLY -- cut line B2 of SURVEY.md 8b).

Registered as ``sys.modules["pycolmap"]`` by ``oracle/gen_golden_triangulator.py`` BEFORE the reference is
imported, it lets the reference's own driver code run unmodified in the build container:
``vggsfm.models.Triangulator.forward`` -> ``init_BA`` / ``init_refine_pose`` / ``global_BA`` /
``iterative_global_BA`` (vggsfm/utils/triangulation.py) -> ``batch_matrix_to_pycolmap`` /
``pycolmap_to_batch_matrix`` (vggsfm/utils/tensor_to_pycolmap.py) -> *this module*.

Only the surface those files touch is provided (last table row):
``Reconstruction, Track, Camera, Rotation3d, Rigid3d, Image, Point2D, ListPoint2D, BundleAdjustmentOptions,
AbsolutePoseRefinementOptions, AbsolutePoseEstimationOptions, bundle_adjustment, pose_refinement,
absolute_pose_estimation``.  The solvers are ``oracle/ba.py`` (the Ceres/COLMAP restatement, PARITY UNPINNED
against the real pycolmap 3.10 -- see oracle/ba_oracle.h); what this harness pins is everything AROUND the
solver: the reference's mask bookkeeping, thresholds schedule, problem construction and result read-back.
``absolute_pose_estimation``: COLMAP's own (adaptive LO-RANSAC on its private RNG) cannot be restated bit for bit.  By
default it returns None, which the reference handles (vggsfm/utils/triangulation.py:434) by keeping the pose.  With
``ESTIMATION["enabled"]`` it is a DETERMINISTIC restatement of what the drop-in computes on the device
(vggsfm_amd/pose.py: absolute_pose_estimation_batch): the 30 quadratically spaced focal-length factors, points normalised
with the scaled camera, a fixed number of P3P minimal samples (oracle/p3p.py) drawn from uniform numbers of a seeded numpy
generator -- the numbers are recorded per frame so that the GPU test replays them --, winner by (inliers, residual sum,
first factor), then RefineAbsolutePose on the RANSAC inliers.  That puts the reference's `force_estimate` branch
(triangulation.py:406-432) into the golden vectors instead of stubbing it out.
"""
import types

import numpy as np

from . import ba as OB


class Track:
    def __init__(self):
        self.elements = []

    def add_element(self, image_id, point2D_idx):
        self.elements.append(types.SimpleNamespace(image_id=int(image_id), point2D_idx=int(point2D_idx)))

    def length(self):
        return len(self.elements)


class Point3D:
    def __init__(self, xyz, track, color):
        self.xyz = np.array(xyz, dtype=np.float64)
        self.track = track
        self.color = np.array(color)
        self.error = -1.0


class Rotation3d:
    def __init__(self, R=None):
        self._R = np.eye(3) if R is None else np.array(R, dtype=np.float64).reshape(3, 3)

    def matrix(self):
        return self._R.copy()


class Rigid3d:
    def __init__(self, rotation=None, translation=None):
        self.rotation = rotation if rotation is not None else Rotation3d()
        self.translation = np.zeros(3) if translation is None else np.array(translation, dtype=np.float64).reshape(3)

    def matrix(self):
        return np.concatenate([self.rotation.matrix(), self.translation[:, None]], axis=1)

    def __mul__(self, xyz):
        return self.rotation.matrix() @ np.asarray(xyz, dtype=np.float64) + self.translation


class Camera:
    def __init__(self, model="SIMPLE_PINHOLE", width=0, height=0, params=(), camera_id=0):
        self.model = str(model)
        self.width, self.height = int(width), int(height)
        self.params = np.array([float(p) for p in params], dtype=np.float64)
        self.camera_id = int(camera_id)

    def calibration_matrix(self):
        f, cx, cy = self.params[0], self.params[1], self.params[2]
        return np.array([[f, 0.0, cx], [0.0, f, cy], [0.0, 0.0, 1.0]])


class Point2D:
    def __init__(self, xy, point3D_id=-1):
        self.xy = np.array(xy, dtype=np.float64)       # pycolmap stores Eigen::Vector2d (float32 tracks widen exactly)
        self.point3D_id = int(point3D_id)


class ListPoint2D(list):
    pass


class Image:
    def __init__(self, id=0, name="", camera_id=0, cam_from_world=None):
        self.image_id = int(id)
        self.name = name
        self.camera_id = int(camera_id)
        self.cam_from_world = cam_from_world if cam_from_world is not None else Rigid3d()
        self.points2D = ListPoint2D()
        self.registered = False


class Reconstruction:
    def __init__(self):
        self.points3D, self.images, self.cameras = {}, {}, {}
        self._next_point3D_id = 1

    def add_point3D(self, xyz, track, color=np.zeros(3)):
        pid = self._next_point3D_id
        self._next_point3D_id += 1
        self.points3D[pid] = Point3D(xyz, track, color)
        return pid

    def add_camera(self, camera):
        self.cameras[camera.camera_id] = camera

    def add_image(self, image):
        self.images[image.image_id] = image

    def point3D_ids(self):
        return set(self.points3D.keys())

    def reg_image_ids(self):
        return sorted(i for i, im in self.images.items() if im.registered)

    def num_points3D(self):
        return len(self.points3D)

    def num_images(self):
        return len(self.images)

    def normalize(self, extent=10.0, p0=0.1, p1=0.9, use_images=True):
        assert use_images
        ids = sorted(self.images)
        ext = np.stack([self.images[i].cam_from_world.matrix() for i in ids])
        pids = sorted(self.points3D)
        pts = np.stack([self.points3D[p].xyz for p in pids]) if pids else np.zeros((0, 3))
        ext, pts = OB.normalize_reconstruction(ext, pts, None, extent, p0, p1)
        for k, i in enumerate(ids):
            self.images[i].cam_from_world = Rigid3d(Rotation3d(ext[k, :, :3]), ext[k, :, 3])
        for k, p in enumerate(pids):
            self.points3D[p].xyz = pts[k]

    # ---- dense view of the problem the reference built with batch_matrix_to_pycolmap
    def _dense(self):
        ids = sorted(self.images)
        assert ids == list(range(len(ids))), "image ids are frame indices (tensor_to_pycolmap.py:118)"
        S = len(ids)
        P = max(self.points3D) if self.points3D else 0
        pts = np.zeros((P, 3))
        alive = np.zeros(P, bool)
        for pid, p in self.points3D.items():
            pts[pid - 1] = p.xyz
            alive[pid - 1] = True
        tracks = np.zeros((S, P, 2), np.float32)
        masks = np.zeros((S, P), bool)
        for i in ids:
            if not self.images[i].registered:
                continue
            for p2 in self.images[i].points2D:
                if p2.point3D_id >= 1 and alive[p2.point3D_id - 1]:
                    tracks[i, p2.point3D_id - 1] = p2.xy
                    masks[i, p2.point3D_id - 1] = True
        ext = np.stack([self.images[i].cam_from_world.matrix() for i in ids])
        cams = [self.cameras[self.images[i].camera_id] for i in ids]
        shared = len(self.cameras) == 1 and S > 1
        model = cams[0].model
        K = np.stack([c.calibration_matrix() for c in cams])
        extra = np.stack([c.params[3:4] for c in cams]) if model == "SIMPLE_RADIAL" else None
        return pts, ext, K, tracks, masks, extra, shared, model


class _SolverOptions:
    def __init__(self):
        # COLMAP 3.10 BundleAdjustmentOptions: ceres defaults overridden in bundle_adjustment.h
        self.function_tolerance = 0.0
        self.gradient_tolerance = 1e-4
        self.parameter_tolerance = 0.0
        self.max_num_iterations = 100
        self.max_linear_solver_iterations = 200
        self.minimizer_progress_to_stdout = False
        self.num_threads = -1


class BundleAdjustmentOptions:
    def __init__(self):
        self.solver_options = _SolverOptions()
        self.refine_focal_length = True
        self.refine_principal_point = False
        self.refine_extra_params = True
        self.refine_extrinsics = True
        self.print_summary = True
        self.loss_function_type = "TRIVIAL"
        self.loss_function_scale = 1.0


class AbsolutePoseRefinementOptions:
    def __init__(self):
        self.gradient_tolerance = 1.0
        self.max_num_iterations = 100
        self.loss_function_scale = 1.0
        self.refine_focal_length = True
        self.refine_extra_params = True
        self.print_summary = False


class _Ransac:
    max_error = 12.0


class AbsolutePoseEstimationOptions:
    def __init__(self):
        self.estimate_focal_length = False
        self.ransac = _Ransac()


CALLS = []      # (kind, summary) of every solver call, for the generator's log


def bundle_adjustment(reconstruction, options=None):
    """pycolmap.bundle_adjustment: all registered images, image 0 pose constant, image 1 t_x constant,
    negative-depth observation filter, Ceres LM (oracle/ba.py::bundle_adjustment)."""
    options = options or BundleAdjustmentOptions()
    pts, ext, K, tracks, masks, extra, shared, model = reconstruction._dense()
    so = options.solver_options
    o = OB.ceres_options(so.max_num_iterations, so.function_tolerance, so.gradient_tolerance, so.parameter_tolerance)
    assert options.refine_focal_length and options.refine_extra_params and not options.refine_principal_point
    p_opt, ext_o, K_o, extra_o, summ = OB.bundle_adjustment(pts, ext, K, tracks, masks, extra, shared, model, options=o)
    CALLS.append(("bundle_adjustment", {k: summ[k] for k in ("initial_cost", "final_cost", "num_iterations",
                                                             "termination")}))
    valid_idx, deleted = summ["valid_idx"], summ["deleted"]
    for k, vi in enumerate(valid_idx):
        pid = int(vi) + 1
        if pid not in reconstruction.points3D:
            continue
        if deleted[k]:
            del reconstruction.points3D[pid]          # ObservationManager deletes the whole point
        else:
            reconstruction.points3D[pid].xyz = p_opt[k].copy()
    for i in sorted(reconstruction.images):
        reconstruction.images[i].cam_from_world = Rigid3d(Rotation3d(ext_o[i, :, :3]), ext_o[i, :, 3])
        cam = reconstruction.cameras[reconstruction.images[i].camera_id]
        cam.params[0], cam.params[1], cam.params[2] = K_o[i, 0, 0], K_o[i, 0, 2], K_o[i, 1, 2]
        if model == "SIMPLE_RADIAL":
            cam.params[3] = extra_o[i, 0]
    return summ


def pose_refinement(cam_from_world, points2D, points3D, inlier_mask, camera, refinement_options=None):
    """pycolmap.pose_refinement -> {"cam_from_world": Rigid3d}; `camera.params` are refined in place."""
    ro = refinement_options or AbsolutePoseRefinementOptions()
    params = np.zeros(4)
    params[:len(camera.params)] = camera.params
    ext, intr, summ = OB.pose_refinement(cam_from_world.matrix(), np.asarray(points2D), np.asarray(points3D),
                                         np.asarray(inlier_mask, bool), params, camera.model,
                                         refine_focal_length=bool(ro.refine_focal_length),
                                         refine_extra_params=bool(ro.refine_extra_params))
    CALLS.append(("pose_refinement", {k: summ[k] for k in ("initial_cost", "final_cost", "num_iterations",
                                                           "termination")}))
    camera.params[:] = intr[:len(camera.params)]
    return {"cam_from_world": Rigid3d(Rotation3d(ext[:, :3]), ext[:, 3]), "num_inliers": int(np.sum(inlier_mask))}


ESTIMATION = {"enabled": False, "rng": None, "num_hypotheses": 1024, "log": []}   # log: (camera_id, n_candidates, uniforms (H,3))


def positions_from_uniforms(r, n):
    """Three DISTINCT positions in [0, n) per row from uniforms r (H,3) -- the arithmetic of vggsfm_amd.pose.draw_minimal_samples."""
    n = max(int(n), 3)
    r0 = np.minimum((r[:, 0] * n).astype(np.int64), n - 1)
    r1 = np.minimum((r[:, 1] * (n - 1)).astype(np.int64), n - 2)
    r1 = r1 + (r1 >= r0)
    r2 = np.minimum((r[:, 2] * (n - 2)).astype(np.int64), n - 3)
    lo, hi = np.minimum(r0, r1), np.maximum(r0, r1)
    r2 = r2 + (r2 >= lo)
    r2 = r2 + (r2 >= hi)
    return np.stack([r0, r1, r2], -1)


def absolute_pose_estimation(points2D, points3D, camera, estimation_options=None, refinement_options=None):
    if not ESTIMATION["enabled"]:
        return None
    from . import geometry as OG
    from . import p3p as OP
    eo = estimation_options or AbsolutePoseEstimationOptions()
    x_px = np.asarray(points2D, np.float64)
    X = np.ascontiguousarray(np.asarray(points3D, np.float64))
    n = len(X)
    if n < 3:
        return None
    H = ESTIMATION["num_hypotheses"]
    r = ESTIMATION["rng"].random((H, 3))
    ESTIMATION["log"].append((int(camera.camera_id), n, r))
    samples = positions_from_uniforms(r, n).astype(np.int64)
    if eo.estimate_focal_length:                           # COLMAP EstimateAbsolutePose: 30 factors in [0.2, 5], quadratic spacing
        factors = np.array([0.2 + (5.0 - 0.2) * (i * (1.0 / 30)) * (i * (1.0 / 30)) for i in range(30)])
    else:
        factors = np.array([1.0])
    G = len(factors)
    f0, cx, cy = float(camera.params[0]), float(camera.params[1]), float(camera.params[2])
    foc = f0 * factors
    K = np.zeros((G, 3, 3))
    K[:, 0, 0] = K[:, 1, 1] = foc
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = cx, cy, 1.0
    extra = np.full((G, 1), float(camera.params[3])) if camera.model == "SIMPLE_RADIAL" else None
    xn = OG.cam_from_img(np.broadcast_to(x_px[None], (G, n, 2)).copy(), K, extra)
    best = None
    mask = np.ones(n, bool)
    for g in range(G):
        res = OP.absolute_pose_ransac(xn[g], X, mask, samples, (float(eo.ransac.max_error) / foc[g]) ** 2)
        if best is None or res["num_inliers"] > best[1]["num_inliers"] or (
                res["num_inliers"] == best[1]["num_inliers"] and res["residual_sum"] < best[1]["residual_sum"]):
            best = (g, res)
    g, res = best
    if res["num_inliers"] < 3:
        return None
    camera.params[0] = foc[g]
    ans = pose_refinement(Rigid3d(Rotation3d(res["pose"][:, :3]), res["pose"][:, 3]), x_px, X, res["inliers"], camera,
                          refinement_options)
    return {"cam_from_world": ans["cam_from_world"], "num_inliers": int(res["num_inliers"]), "inlier_mask": res["inliers"]}


# ------------------------------------------------------------------------------------------------------------------
# The surface ``vggsfm/runners/video_runner.py`` touches on top of the above (move_window :800-838, joint_BA :494-541,
# solve_bundle_adjustment :1321-1331): BundleAdjustmentConfig / BundleAdjuster / pyceres.{SolverSummary, solve} /
# ObservationManager.  Used by oracle/gen_golden_video.py (TEST INFRASTRUCTURE, like everything in this file).
class BundleAdjustmentConfig:
    def __init__(self):
        self.image_ids, self.constant_cam_poses = [], set()
        self.constant_point3D_ids, self.variable_point3D_ids = set(), set()

    def add_image(self, image_id):
        if int(image_id) not in self.image_ids:
            self.image_ids.append(int(image_id))

    def set_constant_cam_pose(self, image_id):
        self.constant_cam_poses.add(int(image_id))

    def add_constant_point(self, point3D_id):
        self.constant_point3D_ids.add(int(point3D_id))

    def add_variable_point(self, point3D_id):
        self.variable_point3D_ids.add(int(point3D_id))


BundleAdjustmentOptions.create_loss_function = lambda self: (self.loss_function_type, self.loss_function_scale)


class BundleAdjuster:
    """pycolmap.BundleAdjuster(options, config): no negative-depth filter (that is the controller's), the gauge is what
    the config says (COLMAP BundleAdjuster::SetUp).  Every image of the reconstruction must be in the config -- the only
    form the reference builds (video_runner.py:817-829)."""

    def __init__(self, options, config):
        self.options, self.config = options, config
        self.reconstruction, self.problem = None, None

    def set_up_problem(self, reconstruction, loss_function=None):
        self.reconstruction = reconstruction
        self.problem = types.SimpleNamespace(adjuster=self)

    def set_up_solver_options(self, problem, solver_options):
        return solver_options

    def _run(self, so):
        rec, cfg, options = self.reconstruction, self.config, self.options
        assert sorted(cfg.image_ids) == rec.reg_image_ids() == sorted(rec.images), "config = all images (video_runner.py:818)"
        pts, ext, K, tracks, masks, extra, shared, model = rec._dense()
        cp = np.zeros(len(pts), bool)
        for pid in cfg.constant_point3D_ids:
            if 1 <= pid <= len(pts):
                cp[pid - 1] = True
        o = OB.ceres_options(so.max_num_iterations, so.function_tolerance, so.gradient_tolerance, so.parameter_tolerance)
        loss = {"TRIVIAL": 0, "CAUCHY": 1, "HUBER": 2, "SOFT_L1": 3}[options.loss_function_type]
        p_opt, ext_o, K_o, extra_o, summ = OB.bundle_adjustment(
            pts, ext, K, tracks, masks, extra, shared, model, options=o, constant_points=cp,
            constant_pose_frames=sorted(cfg.constant_cam_poses), filter_negative_depth=False,
            refine_focal=bool(options.refine_focal_length), refine_extra=bool(options.refine_extra_params), loss=loss,
            loss_scale=float(options.loss_function_scale))
        CALLS.append(("bundle_adjuster", {k: summ[k] for k in ("initial_cost", "final_cost", "num_iterations",
                                                               "termination")}))
        for k, vi in enumerate(summ["valid_idx"]):
            pid = int(vi) + 1
            if pid in rec.points3D:
                rec.points3D[pid].xyz = p_opt[k].copy()
        for i in sorted(rec.images):
            rec.images[i].cam_from_world = Rigid3d(Rotation3d(ext_o[i, :, :3]), ext_o[i, :, 3])
            cam = rec.cameras[rec.images[i].camera_id]
            cam.params[0], cam.params[1], cam.params[2] = K_o[i, 0, 0], K_o[i, 0, 2], K_o[i, 1, 2]
            if model == "SIMPLE_RADIAL":
                cam.params[3] = extra_o[i, 0]
        summ["num_residuals"] = 2 * int(masks[:, summ["valid_idx"]].sum())
        return summ


class _SolverSummary:
    def __init__(self):
        self.num_residuals_reduced = 0
        self.num_effective_parameters_reduced = 0
        self.num_successful_steps = self.num_unsuccessful_steps = 0
        self.total_time_in_seconds = 0.0
        self.initial_cost = self.final_cost = 0.0


def _pyceres_solve(options, problem, summary):
    out = problem.adjuster._run(options)
    summary.num_residuals_reduced = out["num_residuals"]
    summary.num_effective_parameters_reduced = out["n_reduced"]
    summary.num_successful_steps = out["num_successful_steps"]
    summary.num_unsuccessful_steps = out["num_unsuccessful_steps"]
    summary.initial_cost, summary.final_cost = out["initial_cost"], out["final_cost"]
    return summary


pyceres = types.ModuleType("pyceres")      # register as sys.modules["pyceres"] next to this module as "pycolmap"
pyceres.SolverSummary = _SolverSummary
pyceres.solve = _pyceres_solve


class ObservationManager:
    """The two ObservationManager filters of joint_BA (video_runner.py:510-512) [COLMAP 3.10 observation_manager.cc]: an
    observation goes when its squared reprojection error exceeds max^2 or its depth is not positive; a point goes when
    fewer than two observations remain or no pair of its remaining views subtends min_tri_angle degrees."""

    def __init__(self, reconstruction, correspondence_graph=None):
        self.reconstruction = reconstruction

    def _apply(self, inl, keep):
        rec = self.reconstruction
        for i, im in rec.images.items():
            for p2 in im.points2D:
                pid = p2.point3D_id
                if pid >= 1 and pid in rec.points3D and not (keep[pid - 1] and inl[i, pid - 1]):
                    p2.point3D_id = -1
        for pid in [p for p in rec.points3D if not keep[p - 1]]:
            del rec.points3D[pid]
        for pid, p in rec.points3D.items():
            p.track.elements = [e for e in p.track.elements if rec.images[e.image_id].points2D[e.point2D_idx].point3D_id == pid]

    def filter_all_points3D(self, max_reproj_error, min_tri_angle):
        from . import geometry as OG
        pts, ext, K, tracks, masks, extra, shared, model = self.reconstruction._dense()
        tr = tracks.astype(np.float64)
        _, detail = OG.filter_all_points3D(pts, tr, ext, K, extra, max_reproj_error=max_reproj_error, check_triangle=False,
                                           return_detail=True, hard_max=-1)
        inl = masks & detail
        keep, _ = OG.filter_all_points3D(pts, np.where(inl[..., None], tr, 1e9), ext, K, extra,
                                         max_reproj_error=max_reproj_error, min_tri_angle=min_tri_angle,
                                         check_triangle=True, hard_max=-1)
        keep = keep & (inl.sum(0) >= 2)
        before = int(masks.sum())
        self._apply(inl, keep)
        return before - int((inl & keep[None]).sum())

    def filter_observations_with_negative_depth(self):
        pts, ext, K, tracks, masks, extra, shared, model = self.reconstruction._dense()
        z = np.einsum("sj,pj->sp", ext[:, 2, :3], pts) + ext[:, 2, 3][:, None]
        inl = masks & (z >= np.finfo(np.float64).eps)
        keep = inl.sum(0) >= 2                       # (a point left with fewer than two observations is deleted)
        alive = np.zeros(len(pts), bool)
        alive[[p - 1 for p in self.reconstruction.points3D]] = True
        self._apply(inl, keep | ~alive)
        return int(masks.sum()) - int((inl & keep[None]).sum())
