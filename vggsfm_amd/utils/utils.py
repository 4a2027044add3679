"""Camera-prediction averaging of the reference (vggsfm/utils/utils.py:25-187), as torch ops on the device.

``average_camera_prediction`` runs the (learned, injected) camera predictor several times, each time with a different
frame swapped to position 0, brings every prediction back to the frame order and to the gauge of frame 0, and averages:
rotations through the mean of their quaternions, translations and focal lengths arithmetically.  The reference does
the quaternion step on the host with scipy; here it stays on the device and follows scipy's conventions exactly
(``Rotation.from_matrix(...).as_quat()`` of scipy 1.15: SVD orthogonalisation of non-orthogonal input, x,y,z,w,
largest-component branch, no sign canonicalisation) so that the mean
-- which is sign sensitive -- is the reference's.  Used by ``VGGSfMRunner.sparse_reconstruct`` (cfg.avg_pose,
runner.py:392-400) and by the video runner for every window (video_runner.py:662-667): ``VideoGeometry``'s default
``camera_prior`` (``camera_prior_from_predictor``).
"""
import random
import types

import torch


def calculate_index_mappings(query_index, S, device=None):
    """Order that swaps [query_index] and [0] (utils.py:167-177)."""
    new_order = torch.arange(S)
    new_order[0] = query_index
    new_order[query_index] = 0
    return new_order if device is None else new_order.to(device)


def switch_tensor_order(tensors, order, dim=1):
    """utils.py:180-187."""
    return [torch.index_select(t, dim, order) if t is not None else None for t in tensors]


def closed_form_inverse_OpenCV(se3):
    """[R t; 0 1]^-1 = [R^T  -R^T t; 0 1] for a batch of 4x4 matrices (utils/metric.py:233-268)."""
    R, T = se3[:, :3, :3], se3[:, :3, 3:]
    Rt = R.transpose(1, 2)
    inv = torch.eye(4, dtype=se3.dtype, device=se3.device)[None].repeat(len(se3), 1, 1)
    inv[:, :3, :3] = Rt
    inv[:, :3, 3:] = -Rt.bmm(T)
    return inv


def matrix_to_quaternion_scipy(M):
    """(...,3,3) -> (...,4) quaternion (x,y,z,w) with scipy's ``Rotation.from_matrix`` branch structure and sign: the largest
    of (m00, m11, m22, trace) selects the formula; unit norm; the sign is whatever the formula gives."""
    M = M.to(torch.float64)
    # scipy >= 1.11 first orthogonalises an input whose Gramian is not the identity to np.isclose(atol=1e-12) -- every
    # float32 prediction -- by the orthogonal Procrustes solution U V^T of its SVD
    G = M @ M.transpose(-1, -2)
    eye = torch.eye(3, dtype=M.dtype, device=M.device)
    off = ((G - eye).abs() > 1e-12 + 1e-5 * eye).any(-1).any(-1)
    if bool(off.any()):
        U, _, Vh = torch.linalg.svd(M)
        M = torch.where(off[..., None, None], U @ Vh, M)
    m = lambda a, b: M[..., a, b]
    tr = m(0, 0) + m(1, 1) + m(2, 2)
    dec = torch.stack([m(0, 0), m(1, 1), m(2, 2), tr], -1)
    choice = dec.argmax(-1)
    cands = []
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        q = [None] * 4
        q[i] = 1 - dec[..., 3] + 2 * m(i, i)
        q[j] = m(j, i) + m(i, j)
        q[k] = m(k, i) + m(i, k)
        q[3] = m(k, j) - m(j, k)
        cands.append(torch.stack(q, -1))
    cands.append(torch.stack([m(2, 1) - m(1, 2), m(0, 2) - m(2, 0), m(1, 0) - m(0, 1), 1 + dec[..., 3]], -1))
    q = torch.gather(torch.stack(cands, -2), -2, choice[..., None, None].expand(choice.shape + (1, 4))).squeeze(-2)
    return q / q.norm(dim=-1, keepdim=True)


def quaternion_to_matrix_scipy(q):
    """(...,4) unit quaternion (x,y,z,w) -> (...,3,3), scipy's ``Rotation.from_quat(...).as_matrix()`` formula."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    R = torch.stack([x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw),
                     2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw),
                     2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2], -1)
    return R.reshape(q.shape[:-1] + (3, 3))


def average_batch_rotation_matrices(batch_rotation_matrices):
    """(B,N,3,3) -> (N,3,3): normalised mean of the quaternions over B (utils.py:136-164), float64."""
    q = matrix_to_quaternion_scipy(batch_rotation_matrices).mean(0)
    q = q / q.norm(dim=1, keepdim=True)
    # scipy normalises once more inside from_quat
    return quaternion_to_matrix_scipy(q / q.norm(dim=1, keepdim=True))


def average_camera_prediction(camera_predictor, reshaped_image, batch_size, repeat_times=5, query_indices=None):
    """utils.py:25-127.  `camera_predictor(images, batch_size=...)["pred_cameras"]` must expose ``R`` (S,3,3), ``T`` (S,3)
    and ``focal_length`` (S,2) in the OpenCV convention.  Returns a camera object of the predictor's own type when it
    can be built from (focal_length, R, T, device), else a namespace with those three fields."""
    assert batch_size == 1, "This function is designed for inference with batch_size=1."
    num_frames = len(reshaped_image)
    device = reshaped_image.device
    if query_indices is None:
        repeat_times = min(repeat_times, num_frames)
        query_indices = random.sample(range(num_frames), repeat_times)
        if 0 not in query_indices:
            query_indices.insert(0, 0)
    rotations, translations, focal_lengths = [], [], []
    pred_cameras = None
    for query_index in query_indices:
        new_order = calculate_index_mappings(query_index, num_frames, device=device)
        ordered = switch_tensor_order([reshaped_image], new_order, dim=0)[0]
        pred_cameras = camera_predictor(ordered, batch_size=batch_size)["pred_cameras"]
        R, abs_T = pred_cameras.R, pred_cameras.T
        ext = torch.eye(4, dtype=R.dtype, device=R.device)[None].repeat(len(R), 1, 1)
        ext[:, :3, :3] = R
        ext[:, :3, 3] = abs_T
        ext, focal = switch_tensor_order([ext, pred_cameras.focal_length], new_order, dim=0)
        rel = closed_form_inverse_OpenCV(ext[0:1]).expand(len(ext), -1, -1)
        ext = torch.bmm(ext, rel)                     # relative to the first camera (OpenCV convention: right-multiply)
        rotations.append(ext[:, :3, :3][None])
        translations.append(ext[:, :3, 3][None])
        focal_lengths.append(focal[None])
    avg_R = average_batch_rotation_matrices(torch.cat(rotations))
    avg_T = torch.cat(translations).mean(0)
    avg_f = torch.cat(focal_lengths).mean(0)
    try:
        return type(pred_cameras)(focal_length=avg_f, R=avg_R, T=avg_T, device=device)
    except TypeError:
        return types.SimpleNamespace(focal_length=avg_f, R=avg_R, T=avg_T, device=device)


def camera_prior_from_predictor(camera_predictor, images):
    """The ``camera_prior`` callable ``VideoGeometry.move_window`` takes, as the reference builds it
    (video_runner.py:655-681): averaged prediction over the frames [frame_from, frame_to) of `images` (1,S,3,H,W) with
    the query frames (first, middle, last) -> (frame_to - frame_from, 3, 4) extrinsics in the predictor's own gauge."""
    def prior(frame_from, frame_to):
        window = images[:, frame_from:frame_to]
        n = window.shape[1]
        cams = average_camera_prediction(camera_predictor, window.reshape((-1,) + tuple(window.shape[2:])), 1,
                                         query_indices=[0, n // 2, n - 1])
        return torch.cat((cams.R, cams.T.unsqueeze(-1)), dim=-1)
    return prior
