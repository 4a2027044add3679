"""``estimate_preliminary_cameras`` (vggsfm/two_view_geo/estimate_preliminary.py:103-241), the part the pipeline
consumes: the fundamental matrix of every (query frame 0, frame s) pair and its inlier mask -> ``preliminary_dict``
with ``fmat`` / ``fmat_inlier_mask`` / ``fmat_residuals`` (the Triangulator reads ``fmat_inlier_mask``).  The
reference also decomposes an essential matrix into preliminary cameras; its only caller discards them
(vggsfm/runners/runner.py:478 ``_, preliminary_dict = ...``), so they are not produced: the first return value is None,
as with the reference's own poselib variant (estimate_preliminary.py:100)."""
import torch

from .fundamental import estimate_fundamental


def estimate_preliminary_cameras(tracks, tracks_vis, width, height, tracks_score=None, max_error=0.5, lo_num=300,
                                 max_ransac_iters=4096, predict_essential=False, predict_homo=False, loopresidual=False,
                                 samples=None):
    """tracks (B,S,N,2), tracks_vis (B,S,N) [, tracks_score (B,S,N)] -> (None, preliminary_dict)."""
    if predict_essential or predict_homo:
        raise NotImplementedError("essential / homography prediction is not part of the device path")
    B, S, N, _ = tracks.shape
    query = tracks[:, 0:1].expand(-1, S - 1, -1, -1).reshape(B * (S - 1), N, 2)
    ref = tracks[:, 1:].reshape(B * (S - 1), N, 2)
    valid = (tracks_vis >= 0.05)[:, 1:].reshape(B * (S - 1), N)                       # estimate_preliminary.py:118
    if tracks_score is not None:
        valid = valid & (tracks_score >= 0.5)[:, 1:].reshape(B * (S - 1), N)          # :120-126
    fmat, num, mask, res = estimate_fundamental(query, ref, max_ransac_iters=max_ransac_iters, max_error=max_error,
                                                lo_num=lo_num, valid_mask=valid, loopresidual=loopresidual,
                                                return_residuals=True, samples=samples)
    return None, {"fmat": fmat.reshape(B, S - 1, 3, 3), "fmat_inlier_mask": mask.reshape(B, S - 1, N),
                  "fmat_inlier_num": num.reshape(B, S - 1), "fmat_residuals": res.reshape(B, S - 1, N)}
