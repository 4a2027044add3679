"""Edge helpers of the hot path (reference: vggsfm/models/utils.py:38-72, 415-447)."""
import torch
import torch.nn.functional as F

from ..utils.triangulation_helpers import create_intri_matrix


def get_EFP(pred_cameras, image_size, B, S, default_focal=False):
    """Same contract as the reference's vggsfm/models/utils.py:38-72.  `pred_cameras`: anything with .R (B S,3,3),
    .T (B S,3), .focal_length (B S,2) in NDC units of the short image side -> extrinsics (B,S,3,4) and intrinsics
    (B,S,3,3) in pixels with ONE focal length per frame (the mean of fx, fy, kept inside [0.2, 5] x short side; exactly
    the short side with `default_focal`) and the principal point at the image centre."""
    short = image_size.min()
    pose = torch.cat([pred_cameras.R, pred_cameras.T.unsqueeze(-1)], dim=-1).reshape(B, S, 3, 4).clone()
    if default_focal:
        one_f = torch.full((B, S, 1), float(short), dtype=pred_cameras.focal_length.dtype, device=pred_cameras.focal_length.device)
    else:
        f_px = pred_cameras.focal_length.reshape(B, S, 2) * (short / 2)
        one_f = f_px.mean(dim=-1, keepdim=True).clamp(0.2 * short, 5 * short)
    centre = (image_size / 2).to(one_f.dtype).expand(B, S, 2)
    return pose, create_intri_matrix(one_f.expand(B, S, 2), centre)


def sample_features4d(input, coords):
    """Bilinear lookup of (B,C,H,W) features at pixel coords (B,N,2) -> (B,N,C); align_corners=True with the
    reference's pixel convention (vggsfm/models/utils.py:380-447)."""
    B, _, H, W = input.shape
    scale = torch.tensor([2.0 / max(W - 1, 1), 2.0 / max(H - 1, 1)], device=coords.device, dtype=coords.dtype)
    grid = (coords * scale - 1.0).unsqueeze(2)                     # (B,N,1,2)
    feats = F.grid_sample(input, grid.to(input.dtype), align_corners=True, padding_mode="border")
    return feats.permute(0, 2, 3, 1).reshape(B, -1, feats.shape[1])
