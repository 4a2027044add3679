"""Edge helpers of the hot path (reference: vggsfm/models/utils.py:38-72, 415-447)."""
import torch
import torch.nn.functional as F

from ..utils.triangulation_helpers import create_intri_matrix


def get_EFP(pred_cameras, image_size, B, S, default_focal=False):
    """PerspectiveCameras (any object with .R (S,3,3), .T (S,3), .focal_length (S,2) in NDC) ->
    extrinsics (B,S,3,4), intrinsics (B,S,3,3); one-dof focal, principal point at the image centre.
    Reference: vggsfm/models/utils.py:38-72."""
    scale = image_size.min()
    focal_length = pred_cameras.focal_length
    principal_point = torch.zeros_like(focal_length)
    focal_length = focal_length * scale / 2
    principal_point = (image_size[None] - principal_point * scale) / 2
    extrinsics = torch.cat([pred_cameras.R.clone(), pred_cameras.T.clone()[..., None]], dim=-1).reshape(B, S, 3, 4)
    focal_length = focal_length.reshape(B, S, 2)
    principal_point = principal_point.reshape(B, S, 2)
    if default_focal:
        focal_length = torch.full_like(focal_length, float(scale))
    else:
        focal_length = focal_length.mean(dim=-1, keepdim=True).expand(-1, -1, 2)
        focal_length = focal_length.clamp(0.2 * scale, 5 * scale)
    return extrinsics, create_intri_matrix(focal_length, principal_point)


def sample_features4d(input, coords):
    """Bilinear lookup of (B,C,H,W) features at pixel coords (B,N,2) -> (B,N,C); align_corners=True with the
    reference's pixel convention (vggsfm/models/utils.py:380-447)."""
    B, _, H, W = input.shape
    scale = torch.tensor([2.0 / max(W - 1, 1), 2.0 / max(H - 1, 1)], device=coords.device, dtype=coords.dtype)
    grid = (coords * scale - 1.0).unsqueeze(2)                     # (B,N,1,2)
    feats = F.grid_sample(input, grid.to(input.dtype), align_corners=True, padding_mode="border")
    return feats.permute(0, 2, 3, 1).reshape(B, -1, feats.shape[1])
