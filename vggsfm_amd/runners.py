"""The part of ``VGGSfMRunner.sparse_reconstruct`` that follows the tracker (vggsfm/runners/runner.py:467-625), on the
device: two-view stage -> Triangulator -> optional dense extra points -> frame filtering / re-ordering -> back to the
original resolution -> the ``predictions`` dict the rest of the runner (saving, visualisation, dense depth) reads.

The learned parts stay with the caller: ``pred_cameras`` comes from the camera predictor, ``pred_track`` / ``pred_vis``
/ ``pred_score`` from the tracker, and the dense pass asks the caller's tracker for the grid tracks through
``extra_tracker`` (the reference calls ``predict_tracks`` there, runner.py:676-694).  Option names are the reference's
cfg keys (cfgs/demo.yaml)."""
from dataclasses import dataclass

import numpy as np
import torch

from .models import Triangulator
from .models.utils import sample_features4d
from .two_view_geo import estimate_preliminary_cameras
from .utils.triangulation import triangulate_extra_points


@dataclass
class GeometryConfig:
    fmat_thres: float = 4.0
    BA_iters: int = 2
    shared_camera: bool = False
    max_reproj_error: float = 4.0
    init_max_reproj_error: float = 4.0
    extract_color: bool = True
    robust_refine: int = 2
    camera_type: str = "SIMPLE_PINHOLE"
    extra_pt_pixel_interval: int = -1
    extra_by_neighbor: int = -1
    concat_extra_points: bool = False
    filter_invalid_frame: bool = True
    shift_point2d_to_original_res: bool = False
    max_ransac_iters: int = 4096
    lo_num: int = 300


def generate_grid_samples(rect, N=None, pixel_interval=None):
    """vggsfm/utils/utils.py:773-815: (N,2) grid inside rect (1,4) = [x0, y0, x1, y1]; either N points at the rectangle's
    aspect ratio or one every `pixel_interval` pixels (x-major order)."""
    x0, y0, x1, y1 = (float(v) for v in rect[0])
    width, height = x1 - x0, y1 - y0
    if pixel_interval is not None:
        nx, ny = max(1, int(width // pixel_interval)), max(1, int(height // pixel_interval))
    else:
        nx = int(np.sqrt(N * (width / height)))
        ny = int(N / nx)
    gx, gy = torch.meshgrid(torch.linspace(x0, x1, nx, device=rect.device), torch.linspace(y0, y1, ny, device=rect.device),
                            indexing="ij")
    return torch.stack([gx.flatten(), gy.flatten()], dim=-1)


def sample_subrange(N, idx, L):
    """vggsfm/utils/utils.py:818-839: a window of L frames around idx, shifted to stay inside [0, N)."""
    start = idx - L // 2
    end = start + L
    if start < 0:
        end -= start
        start = 0
    if end > N:
        start = max(0, start - (end - N))
        end = N
    if end - start < L:
        if end < N:
            end = min(N, start + L)
        elif start > 0:
            start = max(0, end - L)
    return start, end


def rename_colmap_recons_and_rescale_camera(reconstruction, image_paths, crop_params, img_size,
                                            shift_point2d_to_original_res=False, shared_camera=False):
    """``VGGSfMRunner.rename_colmap_recons_and_rescale_camera`` (runner.py:1009-1054) over the pycolmap object surface:
    images get their file names; cameras go back to the original resolution -- focal x max(real size) / img_size,
    principal point = real size // 2, width / height = real size (only the first camera met when `shared_camera`,
    whose ratio then serves every image, as in the reference); optionally the 2D points too:
    (xy - |crop top-left|) x ratio, one array operation per image.  crop_params (1,S,>=4) tensor:
    [..., :2] real (w, h), [..., -4:-2] crop top-left."""
    cp = crop_params.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(crop_params) else np.asarray(crop_params, np.float64)
    rescale_camera, resize_ratio = True, None
    for image_id in reconstruction.images:
        image = reconstruction.images[image_id]
        camera = reconstruction.cameras[image.camera_id]
        image.name = image_paths[image_id]
        if rescale_camera:
            real = cp[0, image_id, :2]
            resize_ratio = float(real.max() / img_size)
            params = camera.params.copy()
            params[0] = resize_ratio * params[0]
            params[1:3] = real // 2
            camera.params = params
            camera.width, camera.height = real[0], real[1]
        if shift_point2d_to_original_res:
            top_left = np.abs(cp[0, image_id, -4:-2])
            image.points2D._xy[:] = (image.points2D._xy - top_left[None]) * resize_ratio
        if shared_camera:
            rescale_camera = False
    return reconstruction


class GeometryRunner:
    def __init__(self, cfg=None, triangulator=None):
        self.cfg = cfg or GeometryConfig()
        self.triangulator = triangulator or Triangulator()

    # ------------------------------------------------------------------ dense extra points (runner.py:627-742)
    def triangulate_extra_points(self, images, bound_bboxes, intrinsics, extra_params, extrinsics, image_paths, frame_num,
                                 extra_tracker):
        """For every frame: a pixel grid inside its bounding box is tracked through the neighbouring frames by
        `extra_tracker(frame_idx, neighbor_start, neighbor_end, grid_points (1,G,2)) -> (track (1,S',G,2), vis, score)`,
        triangulated with the refined cameras and filtered.  Returns {image_path: {points3D, points3D_rgb, uv}}."""
        cfg = self.cfg
        out = {}
        for frame_idx in range(frame_num):
            rect = bound_bboxes[:, frame_idx].clone().floor()
            rect[:, :2] += cfg.extra_pt_pixel_interval // 2
            rect[:, 2:] -= cfg.extra_pt_pixel_interval // 2
            grid = generate_grid_samples(rect, pixel_interval=cfg.extra_pt_pixel_interval).floor()
            grid_rgb = sample_features4d(images[:, frame_idx], grid[None]).squeeze(0)
            n0, n1 = sample_subrange(frame_num, frame_idx, cfg.extra_by_neighbor) if cfg.extra_by_neighbor > 0 else (0, frame_num)
            track, vis, score = extra_tracker(frame_idx, n0, n1, grid[None])
            ep = None if extra_params is None else extra_params[n0:n1]
            pts, valid = triangulate_extra_points(track.squeeze(0), vis.squeeze(0), score.squeeze(0), extrinsics[n0:n1],
                                                  intrinsics[n0:n1], ep, max_reproj_error=cfg.max_reproj_error)
            out[image_paths[frame_idx]] = {"points3D": pts[valid], "points3D_rgb": grid_rgb[valid], "uv": grid[valid]}
        return out

    # ------------------------------------------------------------------ runner.py:467-625
    def sparse_reconstruct_from_tracks(self, pred_cameras, pred_track, pred_vis, pred_score, images, crop_params=None,
                                       image_paths=None, center_order=None, back_to_original_resolution=True,
                                       extra_tracker=None, bound_bboxes=None, masks=None):
        cfg = self.cfg
        B, S, _, H, W = images.shape
        device = pred_track.device
        predictions = {}
        if image_paths is None:
            image_paths = [f"image_{s}" for s in range(S)]
        _, preliminary_dict = estimate_preliminary_cameras(pred_track, pred_vis, W, H, tracks_score=pred_score, decompose=False,
                                                           max_error=cfg.fmat_thres, loopresidual=True,
                                                           max_ransac_iters=cfg.max_ransac_iters, lo_num=cfg.lo_num)
        (extrinsics_opencv, intrinsics_opencv, extra_params, points3D, points3D_rgb, reconstruction, valid_frame_mask,
         valid_2D_mask, valid_tracks) = self.triangulator(
            pred_cameras, pred_track, pred_vis, images, preliminary_dict, pred_score=pred_score, BA_iters=cfg.BA_iters,
            shared_camera=cfg.shared_camera, max_reproj_error=cfg.max_reproj_error,
            init_max_reproj_error=cfg.init_max_reproj_error, extract_color=cfg.extract_color,
            robust_refine=cfg.robust_refine, camera_type=cfg.camera_type)
        additional_points_dict = None
        if cfg.extra_pt_pixel_interval > 0:
            if extra_tracker is None or bound_bboxes is None:
                raise ValueError("extra_pt_pixel_interval > 0 needs extra_tracker and bound_bboxes")
            additional_points_dict = self.triangulate_extra_points(images, bound_bboxes, intrinsics_opencv, extra_params,
                                                                   extrinsics_opencv, image_paths, S, extra_tracker)
            add_xyz = torch.cat([additional_points_dict[n]["points3D"] for n in image_paths], dim=0)
            add_rgb = torch.cat([additional_points_dict[n]["points3D_rgb"] for n in image_paths], dim=0)
            additional_points_dict["sfm_points_num"] = len(points3D)
            additional_points_dict["additional_points_num"] = len(add_xyz)
            if cfg.concat_extra_points:              # runner.py:549-559, all points at once (empty tracks)
                reconstruction.add_points3D(add_xyz.cpu().numpy(), (add_rgb * 255).long().cpu().numpy())
                points3D = torch.cat([points3D, add_xyz.to(points3D.dtype)], dim=0)
                points3D_rgb = torch.cat([points3D_rgb, add_rgb.to(points3D_rgb.dtype)], dim=0)
        if cfg.filter_invalid_frame:
            extrinsics_opencv = extrinsics_opencv[valid_frame_mask]
            intrinsics_opencv = intrinsics_opencv[valid_frame_mask]
            if extra_params is not None:
                extra_params = extra_params[valid_frame_mask]
            for invalid_id in torch.nonzero(~valid_frame_mask).squeeze(1).cpu().tolist():
                reconstruction.deregister_image(invalid_id)
        img_size = images.shape[-1]
        if center_order is not None:                       # the images were re-ordered around the query frame: undo it
            extrinsics_opencv = extrinsics_opencv[center_order]
            intrinsics_opencv = intrinsics_opencv[center_order]
            if extra_params is not None:
                extra_params = extra_params[center_order]
            pred_track = pred_track[:, center_order]
            pred_vis = pred_vis[:, center_order]
            if pred_score is not None:
                pred_score = pred_score[:, center_order]
        if back_to_original_resolution:
            if crop_params is None:
                raise ValueError("back_to_original_resolution needs crop_params")
            reconstruction = rename_colmap_recons_and_rescale_camera(
                reconstruction, image_paths, crop_params, img_size, shared_camera=cfg.shared_camera,
                shift_point2d_to_original_res=cfg.shift_point2d_to_original_res)
            # intrinsics at the original resolution, in the order of the sorted image paths (runner.py:593-608)
            fname_to_id = {reconstruction.images[i].name: i for i in reconstruction.images}
            K = [reconstruction.cameras[reconstruction.images[fname_to_id[n]].camera_id].calibration_matrix()
                 for n in sorted(image_paths)]
            intrinsics_opencv = torch.from_numpy(np.stack(K)).to(device)
        predictions.update(extrinsics_opencv=extrinsics_opencv, intrinsics_opencv=intrinsics_opencv, points3D=points3D,
                           points3D_rgb=points3D_rgb, reconstruction=reconstruction, extra_params=extra_params,
                           unproj_dense_points3D=None, valid_2D_mask=valid_2D_mask, pred_track=pred_track, pred_vis=pred_vis,
                           pred_score=pred_score, valid_tracks=valid_tracks, additional_points_dict=additional_points_dict,
                           preliminary_dict=preliminary_dict)
        return predictions
