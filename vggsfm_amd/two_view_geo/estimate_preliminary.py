"""``estimate_preliminary_cameras`` (vggsfm/two_view_geo/estimate_preliminary.py:103-241) on the device: the fundamental
matrix of every (query frame 0, frame s) pair and its inlier mask (``vggsfm_amd.two_view_geo.estimate_fundamental``), then
the preliminary cameras the reference derives from it -- default intrinsics (focal = max(width, height), principal point
at the image centre, :244-272), E = K2^T F K1 (fundamental.py:186-212), the four (R, t) candidates of the SVD of E
(essential.py:36-83) and the one with the most points in front of both cameras at a bounded depth (utils.py:325-448;
the two-view triangulations run in the ``vgg_triangulate_by_pair`` kernel, all pairs x 4 candidates in one launch).

Returns ``(pred_cameras, preliminary_dict)`` like the reference: ``pred_cameras`` in the PyTorch3D convention (R transposed,
x / y flipped; a ``minipytorch3d`` ``PerspectiveCameras`` when that package is importable, a namespace with ``R`` / ``T``
otherwise); ``preliminary_dict`` = fmat, fmat_inlier_mask, fmat_residuals, R_opencv, t_opencv, default_intri, emat_fromf
(+ fmat_inlier_num).  The runner keeps only the dict (vggsfm/runners/runner.py:478) and the Triangulator only reads
``fmat_inlier_mask``.  The reference's 5-point essential and 4-point homography RANSACs (essential.py, homography.py)
are not called by this function there either (`predict_essential` / `predict_homo` are dead parameters: "TODO: also clean
the code", :115) and are not provided."""
import types

import torch

from .fundamental import estimate_fundamental


def get_default_intri(width, height, device, dtype, ratio=1.0):
    """utils.py:492-513 -> (focal length (), principal point (2,), K (3,3))."""
    f = max(width, height) * ratio
    K = torch.tensor([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1]], device=device, dtype=dtype)
    return torch.tensor(f, device=device, dtype=dtype), torch.tensor([width / 2, height / 2], device=device, dtype=dtype), K


def build_default_kmat(width, height, B, S, N, device=None, dtype=None):
    """estimate_preliminary.py:244-272 -> kmat1, kmat2 (B(S-1),3,3), fl (B(S-1),4), pp (B(S-1),4)  [:2 left frame, 2: right]."""
    f, p, _ = get_default_intri(width, height, device, dtype)
    fl = (torch.ones((B, S - 1, 4), device=device, dtype=dtype) * f).reshape(B * (S - 1), 4)
    pp = torch.cat([p, p])[None][None].expand(B, S - 1, -1).reshape(B * (S - 1), 4)
    kmat1 = torch.eye(3, device=device, dtype=dtype)[None].repeat(B * (S - 1), 1, 1)
    kmat2 = kmat1.clone()
    kmat1[:, [0, 1], [0, 1]] = fl[:, :2]
    kmat1[:, [0, 1], 2] = pp[:, :2]
    kmat2[:, [0, 1], [0, 1]] = fl[:, 2:]
    kmat2[:, [0, 1], 2] = pp[:, 2:]
    return kmat1, kmat2, fl, pp


def essential_from_fundamental(fmat, kmat1, kmat2):
    """fundamental.py:186-212 (Hartley / Zisserman 9.12): E = K2^T F K1."""
    return kmat2.transpose(-2, -1) @ fmat @ kmat1


def decompose_essential_matrix(E_mat):
    """essential.py:36-83: SVD of E -> the four candidates Rs (B,4,3,3) = [R1, R1, R2, R2], Ts (B,4,3) = [t, -t, t, -t]
    with R1 = U W V^T, R2 = U W^T V^T, t = last column of U (U, V^T made proper rotations first)."""
    E = E_mat.to(torch.float64)
    U, _, Vt = torch.linalg.svd(E)
    flip = torch.ones(3, dtype=E.dtype, device=E.device)
    flip[2] = -1.0
    U = torch.where((torch.det(U) < 0.0)[..., None, None], U * flip[None, None, :], U)           # last column negated
    Vt = torch.where((torch.det(Vt) < 0.0)[..., None, None], Vt * flip[None, :, None], Vt)      # last row negated
    W = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=E.dtype, device=E.device)
    R1, R2 = U @ W @ Vt, U @ W.t() @ Vt
    T = U[..., -1]
    return torch.stack([R1, R1, R2, R2], dim=1), torch.stack([T, -T, T, -T], dim=1)


def remove_cheirality(R, t, points1, points2, focal_length=None, principal_point=None):
    """utils.py:325-363: of the candidates R (B,C,3,3), t (B,C,3) keep, per pair, the one for which most of the N
    two-view triangulations have a depth in (eps, 1000 x baseline) in both cameras.  points1 / points2 (B,N,2) pixels
    (normalised with focal_length / principal_point (B,4)) -> R (B,3,3), t (B,3)."""
    from ..utils.triangulation import triangulate_by_pair
    p1, p2 = points1.to(torch.float64), points2.to(torch.float64)
    if focal_length is not None:
        pp, fl = principal_point.unsqueeze(1).to(torch.float64), focal_length.unsqueeze(1).to(torch.float64)
        p1 = (p1 - pp[..., :2]) / fl[..., :2]
        p2 = (p2 - pp[..., 2:]) / fl[..., 2:]
    B, C = R.shape[0], R.shape[1]
    N = p1.shape[1]
    dev = R.device
    Rf, tf = R.reshape(B * C, 3, 3).to(torch.float64), t.reshape(B * C, 3).to(torch.float64)
    numf = torch.empty(B * C, dtype=torch.long, device=dev)
    eye34 = torch.eye(3, 4, dtype=torch.float64, device=dev)[None]
    eps = torch.finfo(torch.float64).eps

    def count(lo, hi, left, right):
        """candidates lo..hi-1 against ONE set of left points: the kernel triangulates (frame 0, frame v) for v >= 1"""
        ext = torch.cat([eye34, torch.cat([Rf[lo:hi], tf[lo:hi, :, None]], -1)], 0)
        pts, _, _ = triangulate_by_pair(ext[None], torch.cat([left[None], right], 0)[None])
        d1 = pts[..., 2]
        d2 = torch.einsum("vj,vnj->vn", Rf[lo:hi, 2], pts) + tf[lo:hi, 2][:, None]
        max_depth = 1000.0 * torch.linalg.norm(torch.einsum("vji,vj->vi", Rf[lo:hi], tf[lo:hi]), dim=1)[:, None]
        numf[lo:hi] = ((d1 > eps) & (d1 < max_depth) & (d2 > eps) & (d2 < max_depth)).sum(-1)

    if bool((p1 == p1[0:1]).all()):
        # every pair has the same left frame (estimate_preliminary_cameras: the query frame): many pairs per launch
        per = max(C, ((1 << 23) // max(N, 1)) // C * C)
        for lo in range(0, B * C, per):
            hi = min(lo + per, B * C)
            count(lo, hi, p1[0], p2[torch.arange(lo, hi, device=dev) // C])
    else:
        for pr in range(B):
            count(pr * C, (pr + 1) * C, p1[pr], p2[pr][None].expand(C, -1, -1))
    nums = numf.reshape(B, C)
    idx = torch.argmax(nums, dim=1)
    ar = torch.arange(B, device=R.device)
    return R[ar, idx], t[ar, idx]


def _pytorch3d_cameras(R_opencv, t_opencv):
    """OpenCV / COLMAP -> PyTorch3D convention (estimate_preliminary.py:199-222); the relative-to-first step that follows
    there is the identity, the first camera being the identity."""
    R = R_opencv.clone().permute(0, 2, 1)
    T = t_opencv.clone()
    T[:, :2] *= -1
    R[:, :, :2] *= -1
    try:
        from minipytorch3d.cameras import PerspectiveCameras
        return PerspectiveCameras(R=R, T=T, device=R.device)
    except Exception:
        return types.SimpleNamespace(R=R, T=T, device=R.device)


def estimate_preliminary_cameras(tracks, tracks_vis, width, height, tracks_score=None, max_error=0.5, lo_num=300,
                                 max_ransac_iters=4096, predict_essential=False, predict_homo=False, loopresidual=False,
                                 samples=None, decompose=True):
    """tracks (B,S,N,2), tracks_vis (B,S,N) [, tracks_score (B,S,N)] -> (pred_cameras, preliminary_dict).
    decompose=False: only the fundamental-matrix part of the dictionary (`fmat*`) and pred_cameras = None -- the runners
    discard the cameras (runner.py:487-499 keeps `fmat_inlier_mask` alone), and the essential-matrix SVD, the four
    cheirality triangulations and their host synchronisation are then not paid for."""
    B, S, N, _ = tracks.shape
    query = tracks[:, 0:1].expand(-1, S - 1, -1, -1).reshape(B * (S - 1), N, 2)
    ref = tracks[:, 1:].reshape(B * (S - 1), N, 2)
    valid = (tracks_vis >= 0.05)[:, 1:].reshape(B * (S - 1), N)                       # estimate_preliminary.py:118
    if tracks_score is not None:
        valid = valid & (tracks_score >= 0.5)[:, 1:].reshape(B * (S - 1), N)          # :120-126
    fmat, num, mask, res = estimate_fundamental(query, ref, max_ransac_iters=max_ransac_iters, max_error=max_error,
                                                lo_num=lo_num, valid_mask=valid, loopresidual=loopresidual,
                                                return_residuals=True, samples=samples)
    out = {"fmat": fmat.reshape(B, S - 1, 3, 3), "fmat_inlier_mask": mask.reshape(B, S - 1, N),
           "fmat_inlier_num": num.reshape(B, S - 1), "fmat_residuals": res.reshape(B, S - 1, N)}
    if not decompose:
        return None, out
    kmat1, kmat2, fl, pp = build_default_kmat(width, height, B, S, N, device=tracks.device, dtype=torch.float64)
    emat = essential_from_fundamental(fmat.to(torch.float64), kmat1, kmat2)
    Rs, Ts = decompose_essential_matrix(emat)
    R_sel, t_sel = remove_cheirality(Rs, Ts, query, ref, fl, pp)
    eye = torch.eye(3, dtype=torch.float64, device=tracks.device)[None].repeat(B, 1, 1).unsqueeze(1)
    R_opencv = torch.cat([eye, R_sel.reshape(B, S - 1, 3, 3)], dim=1)
    t_opencv = torch.cat([torch.zeros((B, 1, 3), dtype=torch.float64, device=tracks.device), t_sel.reshape(B, S - 1, 3)], dim=1)
    pred_cameras = _pytorch3d_cameras(R_opencv.reshape(B * S, 3, 3), t_opencv.reshape(B * S, 3))
    out.update({"R_opencv": R_opencv, "t_opencv": t_opencv, "default_intri": kmat1.reshape(B, S - 1, 3, 3), "emat_fromf": emat})
    return pred_cameras, out
