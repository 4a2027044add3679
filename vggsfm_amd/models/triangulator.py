"""Drop-in for ``vggsfm.models.Triangulator`` (reference vggsfm/models/triangulator.py:30-476) on the
MI355X kernels.  Same constructor, same ``forward`` signature and the same 9-tuple; select it through the
hydra target (``MODEL.triangulator._target_: vggsfm_amd.models.Triangulator``).

The driver logic (initial pair -> init BA -> pose refinement -> LO-RANSAC triangulation -> global BA ->
robust refine -> iterative BA -> masks) is restated from the reference; every heavy step is one of the
device functions in ``vggsfm_amd.utils``.
"""
import numpy as np
import torch
import torch.nn as nn

from ..utils.tensor_to_pycolmap import batch_matrix_to_pycolmap
from ..ba_options import BundleAdjustmentOptions
from ..utils.triangulation import (global_BA, init_BA, init_refine_pose, iterative_global_BA, refine_pose,
                                   triangulate_by_pair, triangulate_tracks)
from ..utils.triangulation_helpers import cam_from_img, filter_all_points3D
from .utils import get_EFP, sample_features4d


class Triangulator(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = cfg

    def forward(self, pred_cameras, pred_tracks, pred_vis, images, preliminary_dict, pred_score=None,
                init_max_reproj_error=0.5, BA_iters=2, shared_camera=False, max_reproj_error=4, init_tri_angle_thres=16,
                min_valid_track_length=3, robust_refine=2, extract_color=True, camera_type="SIMPLE_PINHOLE"):
        """Reference: triangulator.py:44-363."""
        device = pred_tracks.device
        B, S, _, H, W = images.shape
        _, _, N, _ = pred_tracks.shape
        assert B == 1
        image_size = torch.tensor([W, H], dtype=pred_tracks.dtype, device=device)
        extrinsics, intrinsics = get_EFP(pred_cameras, image_size, B, S)
        extrinsics = extrinsics.double()
        inlier_fmat = preliminary_dict["fmat_inlier_mask"]
        extrinsics, intrinsics = extrinsics[0], intrinsics[0]
        pred_tracks, pred_vis, inlier_fmat = pred_tracks[0], pred_vis[0], inlier_fmat[0]
        pred_score = pred_score[0] if pred_score is not None else torch.ones_like(pred_vis)
        if shared_camera:
            intrinsics[:, 0, 0] = intrinsics[:, 0, 0].mean()
            intrinsics[:, 1, 1] = intrinsics[:, 1, 1].mean()
        extra_params = None
        if camera_type == "SIMPLE_RADIAL":
            extra_params = torch.zeros_like(extrinsics[:, 0, 0:1])

        tracks_normalized = cam_from_img(pred_tracks, intrinsics)
        inlier_geo_vis = torch.logical_and(inlier_fmat, (pred_vis > 0.05)[1:])
        points_3d_pair, cheirality_mask_pair, triangle_value_pair = triangulate_by_pair(extrinsics[None],
                                                                                       tracks_normalized[None])
        inlier_total, _ = find_best_initial_pair(inlier_geo_vis, cheirality_mask_pair, triangle_value_pair,
                                                 init_tri_angle_thres)
        (points3D_init, extrinsics, intrinsics, extra_params, track_init_mask, _, init_idx) = init_BA(
            extrinsics, intrinsics, extra_params, pred_tracks, points_3d_pair, inlier_total, image_size,
            shared_camera=shared_camera, init_max_reproj_error=init_max_reproj_error, camera_type=camera_type)
        print("Finished init BA")
        extrinsics, intrinsics, extra_params, _ = init_refine_pose(
            extrinsics, intrinsics, extra_params, inlier_geo_vis, points3D_init, pred_tracks, track_init_mask, image_size,
            init_idx, shared_camera=shared_camera, camera_type=camera_type)
        print("Finished init refine pose")
        points3D, extrinsics, intrinsics, extra_params, valid_tracks = self.triangulate_tracks_and_BA(
            pred_tracks, intrinsics, extrinsics, extra_params, pred_vis, pred_score, image_size, min_valid_track_length,
            max_reproj_error, shared_camera=shared_camera, camera_type=camera_type)
        print("Finished track triangulation and BA")
        for refine_idx in range(robust_refine):
            extrinsics, intrinsics, extra_params, _ = refine_pose(
                extrinsics, intrinsics, extra_params, pred_vis > 0.05, points3D, pred_tracks, valid_tracks, image_size,
                force_estimate=(refine_idx == robust_refine - 1), shared_camera=shared_camera, camera_type=camera_type)
            points3D, extrinsics, intrinsics, extra_params, valid_tracks = self.triangulate_tracks_and_BA(
                pred_tracks, intrinsics, extrinsics, extra_params, pred_vis, pred_score, image_size, min_valid_track_length,
                max_reproj_error, shared_camera=shared_camera, camera_type=camera_type)
            print(f"Finished robust refine {refine_idx}")

        ba_options = BundleAdjustmentOptions()
        print(f"Running iterative BA by {BA_iters} times")
        BA_inlier_masks = None
        for BA_iter in range(BA_iters):
            lastBA = BA_iter == BA_iters - 1
            ba_options.print_summary = lastBA
            (points3D, extrinsics, intrinsics, extra_params, valid_tracks, BA_inlier_masks, _) = iterative_global_BA(
                pred_tracks, intrinsics, extrinsics, pred_vis, pred_score, valid_tracks, points3D, image_size,
                lastBA=lastBA, extra_params=extra_params, shared_camera=shared_camera,
                min_valid_track_length=min_valid_track_length, max_reproj_error=max_reproj_error, ba_options=ba_options,
                camera_type=camera_type)
            print(f"Finished iterative BA {BA_iter}")
            max_reproj_error = max(max_reproj_error // 2, 1)

        scale = image_size.max()
        valid_param_mask = torch.logical_and(intrinsics[:, 0, 0] >= 0.1 * scale, intrinsics[:, 0, 0] <= 30 * scale)
        if extra_params is not None:
            valid_param_mask = torch.logical_and(valid_param_mask, (extra_params.abs() <= 1.0).all(-1))
        valid_frame_mask = torch.logical_and(valid_param_mask, (extrinsics[:, :3, 3].abs() <= 30).all(-1))
        valid_2D_mask = torch.ones_like(pred_tracks[..., 0]).bool()
        valid_2D_mask[:, ~valid_tracks] = False
        valid_2D_mask[:, valid_tracks] = BA_inlier_masks

        reconstruction, points3D_rgb = build_reconstruction(
            points3D, extrinsics, intrinsics, extra_params, pred_tracks, valid_tracks, BA_inlier_masks, image_size,
            images if extract_color else None, shared_camera=shared_camera, camera_type=camera_type)
        return (extrinsics, intrinsics, extra_params, points3D, points3D_rgb, reconstruction, valid_frame_mask,
                valid_2D_mask, valid_tracks)

    def triangulate_tracks_and_BA(self, pred_tracks, intrinsics, extrinsics, extra_params, pred_vis, pred_score, image_size,
                                  min_valid_track_length, max_reproj_error=4, shared_camera=False,
                                  camera_type="SIMPLE_PINHOLE"):
        """Reference: triangulator.py:365-439."""
        tn = cam_from_img(pred_tracks, intrinsics, extra_params)
        best_pts, best_num, best_mask = triangulate_tracks(extrinsics, tn, track_vis=pred_vis, track_score=pred_score)
        valid_tracks = best_num >= min_valid_track_length
        points3D, extrinsics, intrinsics, extra_params, _ = global_BA(
            best_pts, valid_tracks, pred_tracks, best_mask, extrinsics, intrinsics, extra_params, image_size,
            shared_camera=shared_camera, camera_type=camera_type)
        valid3D, _ = filter_all_points3D(points3D, pred_tracks[:, valid_tracks], extrinsics, intrinsics, extra_params,
                                         check_triangle=False, max_reproj_error=max_reproj_error)
        points3D = points3D[valid3D]
        vt = valid_tracks.clone()
        vt[valid_tracks] = valid3D
        return points3D, extrinsics, intrinsics, extra_params, vt


def build_reconstruction(points3D, extrinsics, intrinsics, extra_params, pred_tracks, valid_tracks, BA_inlier_masks,
                         image_size, images=None, shared_camera=False, camera_type="SIMPLE_PINHOLE"):
    """The model the runner edits and saves, with the pycolmap object surface (``vggsfm_amd.pycolmap_compat``):
    ``batch_matrix_to_pycolmap`` over the final tracks, then ``filter_reconstruction`` = ``normalize(5, 0.1, 0.9, True)``
    (triangulation.py:1187-1218); point colours = mean colour of the inlier observations (triangulator.py:320-342;
    `images` (1,S,3,H,W) or None).  Returns (reconstruction, points3D_rgb (P,3) | None)."""
    reconstruction = batch_matrix_to_pycolmap(points3D, extrinsics, intrinsics, pred_tracks[:, valid_tracks],
                                              BA_inlier_masks, image_size, shared_camera=shared_camera,
                                              camera_type=camera_type, extra_params=extra_params)
    reconstruction.normalize(5.0, 0.1, 0.9, True)
    points3D_rgb = None
    if images is not None:
        pred_track_rgb = sample_features4d(images.squeeze(0), pred_tracks)
        valid_track_rgb = pred_track_rgb[:, valid_tracks]
        sum_rgb = (BA_inlier_masks.float()[..., None] * valid_track_rgb).sum(dim=0)
        points3D_rgb = sum_rgb / BA_inlier_masks.sum(dim=0)[:, None]
        if points3D_rgb.shape[0] == max(reconstruction.point3D_ids()):
            reconstruction.set_colors(np.round(points3D_rgb.cpu().numpy() * 255).astype(np.uint8))
        else:
            print("Cannot save point rgb colors to colmap reconstruction object.")
    return reconstruction, points3D_rgb


def find_best_initial_pair(inlier_geo_vis, cheirality_mask_pair, triangle_value_pair, init_tri_angle_thres):
    """Same contract as the reference's triangulator.py:442-476: relax the triangulation-angle threshold by integer
    halving until the best frame pair keeps >= 100 inliers that are >= a quarter of all tracks -- at most five halvings,
    and none once the threshold is below 2.  Returns the pair-inlier mask of the LAST threshold that was evaluated and the
    threshold after the last halving (after five unsuccessful rounds those differ by one halving, as in the reference)."""
    n_tracks = inlier_geo_vis.shape[-1]
    eligible = inlier_geo_vis & cheirality_mask_pair
    thres = init_tri_angle_thres
    for _ in range(5):
        candidates = eligible & (triangle_value_pair >= thres)
        best = int(candidates.sum(dim=-1).max())
        if (best >= 100 and 4 * best >= n_tracks) or thres < 2:
            break
        thres = thres // 2
    return candidates, thres
