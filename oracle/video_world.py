"""A seeded synthetic world for the video path (TEST INFRASTRUCTURE): ground-truth trajectory + point cloud and the two
learned parts of ``VideoRunner`` -- camera predictor and tracker -- as deterministic providers.

Both sides of the video golden use the SAME world and make the SAME sequence of provider calls:
  * oracle/gen_golden_video.py binds the providers to the names the reference's ``video_runner`` module calls
    (``average_camera_prediction``, ``predict_tracks``) and runs the reference's UNMODIFIED ``VideoRunner.run`` loop;
  * tests/test_gpu_video_golden.py hands them to ``vggsfm_amd.video.VideoGeometry`` as ``camera_prior`` /
    ``track_existing`` / ``track_new``.
The world is regenerated from its seed on either side (a sha256 of it is kept in the golden file, the pattern of the
compact Triangulator goldens).  numpy only.

Provider behaviour:
  camera_prior(f0, f1)        noisy ground truth in a fresh similarity gauge per call (sequential rng)
  track_existing(f0, f1, uv)  the query pixel identifies the ground-truth point (exact float32 match in frame f0; queries
                              that match nothing -- the reference's random support points -- come back as junk); scripted
                              OCCLUSIONS {call number: first window index whose visibility is zero} drive the reference's
                              shrink and step-back branches (video_runner.py:712-751)
  track_new(f0, f1, qframes)  `n_new` ground-truth points visible in every query frame (sequential rng), projections +
                              fixed pixel noise, pixels outside the image are junk with visibility 0
"""
import hashlib

import numpy as np


def rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


class VideoWorld:
    JUNK = -4000.0

    def __init__(self, T, N, seed, n_new=400, occlusions=None, W=1024.0, f=1000.0, k1=0.02, noise_px=0.3):
        rng = np.random.default_rng(seed)
        self.T, self.N, self.W, self.f, self.k1, self.n_new = T, N, W, f, k1, n_new
        ext = np.zeros((T, 3, 4))
        for t in range(T):
            R = rodrigues(np.array([0.02 * np.sin(0.3 * t), 0.06 * np.sin(0.15 * t), 0.01 * np.cos(0.2 * t)]))
            c = np.array([0.11 * t, 0.05 * np.sin(0.25 * t), 0.03 * np.cos(0.2 * t)])
            ext[t, :, :3], ext[t, :, 3] = R, -R @ c
        pts = np.stack([rng.uniform(-2.5, 0.11 * T + 2.5, N), rng.uniform(-1.5, 1.5, N), rng.uniform(3.0, 6.0, N)], 1)
        Xc = np.einsum("tij,nj->tni", ext[:, :, :3], pts) + ext[:, None, :, 3]
        u, v = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
        d = 1.0 + k1 * (u * u + v * v)
        uv = np.stack([f * u * d + W / 2, f * v * d + W / 2], -1)
        vis = (uv[..., 0] > 8) & (uv[..., 0] < W - 8) & (uv[..., 1] > 8) & (uv[..., 1] < W - 8) & (Xc[..., 2] > 0.5)
        uv = uv + rng.normal(0, noise_px, uv.shape)
        uv[~vis] = self.JUNK
        self.ext, self.pts, self.vis = ext, pts, vis
        self.tracks = uv.astype(np.float32)
        self.K = np.array([[f, 0, W / 2], [0, f, W / 2], [0, 0, 1.0]])
        self.prior_rng = np.random.default_rng(seed + 1)
        self.new_rng = np.random.default_rng(seed + 2)
        self.init_rng = np.random.default_rng(seed + 3)
        self.occlusions = dict(occlusions or {})
        self.existing_calls = 0
        self.log = []                                     # (kind, f0, f1, detail) of every provider call, in order

    def digest(self):
        h = hashlib.sha256()
        for a in (self.ext, self.pts, self.tracks, self.vis):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    # ------------------------------------------------------------------ what sparse_reconstruct hands over
    def initial_prediction(self, init, n_points=1200, min_views=4, xyz_noise=0.01):
        """A perturbed reconstruction of frames 0..init (the dict ``convert_pred_to_point_frame_dict`` consumes)."""
        seen = self.vis[:init].sum(0) >= min_views
        idx = np.nonzero(seen)[0][:n_points]
        return dict(extrinsics=self.ext[:init].copy(), tracks=self.tracks[:init][:, idx].copy(),
                    vis=self.vis[:init][:, idx].astype(np.float32), mask=self.vis[:init][:, idx].copy(),
                    points3D=self.pts[idx] + self.init_rng.normal(0, xyz_noise, (len(idx), 3)))

    # ------------------------------------------------------------------ camera predictor
    def camera_prior(self, f0, f1):
        e = self.ext[f0:f1]
        rng = self.prior_rng
        Rg, s, tg = rodrigues(rng.normal(0, 0.5, 3)), float(rng.uniform(0.5, 2.0)), rng.normal(0, 1.0, 3)
        out = np.zeros_like(e)
        for i in range(len(e)):
            Rn = rodrigues(rng.normal(0, 0.01, 3)) @ e[i, :, :3]
            tn = e[i, :, 3] + rng.normal(0, 0.02, 3)
            out[i, :, :3] = Rn @ Rg.T                                  # X_world' = s Rg X + tg
            out[i, :, 3] = s * tn - out[i, :, :3] @ tg
        self.log.append(("camera_prior", f0, f1, None))
        return out

    # ------------------------------------------------------------------ tracker
    def _match(self, f0, uv):
        """ground-truth id of every query pixel of frame f0 (-1: no such observation), by exact float32 equality"""
        uv = np.ascontiguousarray(np.asarray(uv, np.float32).reshape(-1, 2))
        cand = np.nonzero(self.vis[f0])[0]
        table = {self.tracks[f0, i].tobytes(): int(i) for i in cand}
        return np.array([table.get(uv[q].tobytes(), -1) for q in range(len(uv))], np.int64)

    def track_existing(self, f0, f1, uv):
        ids = self._match(f0, uv)
        ok = ids >= 0
        S, P = f1 - f0, len(ids)
        tracks = np.full((S, P, 2), self.JUNK, np.float32)
        vis = np.zeros((S, P), np.float32)
        tracks[:, ok] = self.tracks[f0:f1][:, ids[ok]]
        vis[:, ok] = self.vis[f0:f1][:, ids[ok]]
        first_bad = self.occlusions.get(self.existing_calls)
        if first_bad is not None:
            vis[first_bad:] = 0.0                                      # the tracker loses every query from there on
        self.log.append(("track_existing", f0, f1, (self.existing_calls, int(ok.sum()), first_bad)))   # (matched queries: the reference adds support points)
        self.existing_calls += 1
        return tracks, vis

    def track_new(self, f0, f1, query_frames):
        sel = []
        for q in query_frames:
            cand = np.nonzero(self.vis[f0 + q])[0]
            sel.append(np.sort(self.new_rng.choice(cand, size=min(self.n_new, len(cand)), replace=False)))
        ids = np.concatenate(sel)
        self.log.append(("track_new", f0, f1, (tuple(int(q) for q in query_frames), len(ids))))
        tr = self.tracks[f0:f1][:, ids].copy()
        vis = self.vis[f0:f1][:, ids].astype(np.float32)
        return tr, vis, np.ones_like(vis)
