"""Device-resident track tables for the video path: the tensor replacement of ``VideoRunner.point_dict`` /
``frame_dict`` (vggsfm/runners/video_runner.py:354-473, 543-638), SURVEY.md 8(f).2.

The reference keeps ``point_dict[id] = {xyz, rgb, track: {frame: {uv, vis}}}`` and
``frame_dict[frame] = {extri, visible_points: [ids]}`` as nested Python dicts on the CPU and walks them with
O(points x frames) interpreter loops every window (its authors flag one of them as "too slow",
video_runner.py:194).  Here the same state is a handful of tensors on the device:

    xyz (P,3) f64, rgb (P,3) f32       point table; the point id is the row index (ids are dense, like the dicts' keys)
    obs_point, obs_frame (O,) i64      one row per (point, frame) observation, kept sorted by (point, frame)
    obs_uv (O,2) f32, obs_vis (O,) f32
    extri (F,3,4) f64, has_extri (F,)  per-frame extrinsics

Every method is a few vectorised tensor ops; the dense (tracks, masks) views that bundle adjustment consumes are
produced by scatter, not by loops.
"""
import torch


class TrackTable:
    def __init__(self, device="cuda", num_frames=0):
        self.device = torch.device(device)
        self.xyz = torch.zeros((0, 3), dtype=torch.float64, device=self.device)   # (BA output precision is kept)
        self.rgb = torch.zeros((0, 3), dtype=torch.float32, device=self.device)
        self.obs_point = torch.zeros(0, dtype=torch.long, device=self.device)
        self.obs_frame = torch.zeros(0, dtype=torch.long, device=self.device)
        self.obs_uv = torch.zeros((0, 2), dtype=torch.float32, device=self.device)
        self.obs_vis = torch.zeros(0, dtype=torch.float32, device=self.device)
        self.extri = torch.zeros((num_frames, 3, 4), dtype=torch.float64, device=self.device)
        self.has_extri = torch.zeros(num_frames, dtype=torch.bool, device=self.device)

    # ------------------------------------------------------------------ sizes
    @property
    def num_points(self):
        return int(self.xyz.shape[0])

    @property
    def num_observations(self):
        return int(self.obs_point.shape[0])

    def _grow_frames(self, n):
        if n > self.extri.shape[0]:
            pad = n - self.extri.shape[0]
            self.extri = torch.cat([self.extri, torch.zeros((pad, 3, 4), dtype=self.extri.dtype, device=self.device)])
            self.has_extri = torch.cat([self.has_extri, torch.zeros(pad, dtype=torch.bool, device=self.device)])

    # ------------------------------------------------------------------ updates
    def set_extrinsics(self, start_idx, extrinsics):
        """frame_dict[frame]["extri"] = extrinsics[frame - start_idx] (video_runner.py:382-386)."""
        n = extrinsics.shape[0]
        self._grow_frames(start_idx + n)
        self.extri[start_idx:start_idx + n] = extrinsics.to(self.device, self.extri.dtype)
        self.has_extri[start_idx:start_idx + n] = True

    def _merge_observations(self, point, frame, uv, vis):
        """Insert / overwrite (point, frame) rows (dict assignment semantics: the last write wins) and keep the
        table sorted by (point, frame)."""
        F = max(int(self.extri.shape[0]), int(frame.max().item()) + 1 if frame.numel() else 0, 1)
        p = torch.cat([self.obs_point, point])
        f = torch.cat([self.obs_frame, frame])
        u = torch.cat([self.obs_uv, uv.to(self.device, torch.float32)])
        v = torch.cat([self.obs_vis, vis.to(self.device, torch.float32)])
        key = p * F + f
        order = torch.argsort(key, stable=True)                 # stable: later writes stay behind earlier ones
        key, p, f, u, v = key[order], p[order], f[order], u[order], v[order]
        last = torch.ones_like(key, dtype=torch.bool)
        last[:-1] = key[1:] != key[:-1]                          # keep the last row of every (point, frame) run
        self.obs_point, self.obs_frame, self.obs_uv, self.obs_vis = p[last], f[last], u[last], v[last]

    def update_points(self, start_idx, end_idx, valid_2D_mask, pred_track, pred_vis, points3D_idx=None,
                      point_to_track_mapping=None, points3D=None, points3D_rgb=None, existing_max_point_idx=0):
        """``VideoRunner._update_points_to_dict`` (video_runner.py:407-473).

        valid_2D_mask / pred_track / pred_vis: (S, N[,2]) over the window frames start_idx..end_idx-1;
        points3D_idx (n,) local point ids (default arange); point_to_track_mapping (n,) track column of every local
        point (default identity); absolute id = local id + existing_max_point_idx.  Points that do not exist yet are
        created from points3D / points3D_rgb; existing points only receive the new observations."""
        dev = self.device
        S = end_idx - start_idx
        valid_2D_mask = valid_2D_mask.to(dev).bool()
        pred_track = pred_track.to(dev)
        pred_vis = pred_vis.to(dev)
        if points3D_idx is None:
            points3D_idx = torch.arange(valid_2D_mask.shape[1] if point_to_track_mapping is None
                                        else len(point_to_track_mapping), device=dev)
        points3D_idx = torch.as_tensor(points3D_idx, dtype=torch.long, device=dev)
        if point_to_track_mapping is None:
            point_to_track_mapping = torch.arange(int(points3D_idx.max().item()) + 1 if points3D_idx.numel() else 0,
                                                  device=dev)
        mapping = torch.as_tensor(point_to_track_mapping, dtype=torch.long, device=dev)
        track_idx = mapping[points3D_idx]                         # (n,)
        abs_idx = points3D_idx + existing_max_point_idx
        # create the points that are new (ids are dense: new ids extend the table)
        need = int(abs_idx.max().item()) + 1 if abs_idx.numel() else 0
        if need > self.num_points:
            is_new = abs_idx >= self.num_points
            grow = need - self.num_points
            new_xyz = torch.zeros((grow, 3), dtype=torch.float64, device=dev)
            new_rgb = torch.zeros((grow, 3), dtype=torch.float32, device=dev)
            if points3D is not None:
                new_xyz[abs_idx[is_new] - self.num_points] = points3D.to(dev, torch.float64)[points3D_idx[is_new]]
            if points3D_rgb is not None:
                new_rgb[abs_idx[is_new] - self.num_points] = points3D_rgb.to(dev, torch.float32)[points3D_idx[is_new]]
            self.xyz = torch.cat([self.xyz, new_xyz])
            self.rgb = torch.cat([self.rgb, new_rgb])
        m = valid_2D_mask[:S][:, track_idx]                        # (S, n)
        fr, pi = torch.nonzero(m, as_tuple=True)
        self._grow_frames(end_idx)
        self._merge_observations(abs_idx[pi], fr + start_idx, pred_track[fr, track_idx[pi]], pred_vis[fr, track_idx[pi]])

    def add_window_prediction(self, pred, start_idx, end_idx):
        """``VideoRunner.convert_pred_to_point_frame_dict`` (video_runner.py:354-405)."""
        valid_tracks = pred["valid_tracks"].to(self.device)
        mapping = torch.nonzero(valid_tracks).squeeze(1)
        points3D = pred["points3D"]
        idx = pred.get("points3D_idx")
        if idx is None:
            idx = torch.arange(points3D.shape[0], device=self.device)
        self.set_extrinsics(start_idx, pred["extrinsics_opencv"])
        self.update_points(start_idx, end_idx, pred["valid_2D_mask"], pred["pred_track"], pred["pred_vis"], idx, mapping,
                           points3D, pred.get("points3D_rgb"), existing_max_point_idx=self.num_points)

    # ------------------------------------------------------------------ queries
    def visible_points(self, frame_idx):
        """Sorted ids of the points observed in `frame_idx` (the reference keeps them in insertion order; every
        consumer treats the list as a set)."""
        return self.obs_point[self.obs_frame == frame_idx]

    def track_of(self, point_id):
        sel = self.obs_point == point_id
        return self.obs_frame[sel], self.obs_uv[sel], self.obs_vis[sel]

    def window_tensors(self, start_idx, end_idx, point_ids=None):
        """Dense view for bundle adjustment (what ``dicts_to_reconstruction`` + ``batch_matrix_to_pycolmap`` encode,
        video_runner.py:543-605): (points3D (P',3), extrinsics (S,3,4), tracks (S,P',2), masks (S,P'), point_ids)."""
        S = end_idx - start_idx
        if point_ids is None:
            point_ids = torch.arange(self.num_points, device=self.device)
        point_ids = torch.as_tensor(point_ids, dtype=torch.long, device=self.device)
        col = torch.full((self.num_points,), -1, dtype=torch.long, device=self.device)
        col[point_ids] = torch.arange(point_ids.shape[0], device=self.device)
        sel = (self.obs_frame >= start_idx) & (self.obs_frame < end_idx) & (col[self.obs_point] >= 0)
        tracks = torch.zeros((S, point_ids.shape[0], 2), dtype=torch.float32, device=self.device)
        masks = torch.zeros((S, point_ids.shape[0]), dtype=torch.bool, device=self.device)
        f, c = self.obs_frame[sel] - start_idx, col[self.obs_point[sel]]
        tracks[f, c] = self.obs_uv[sel]
        masks[f, c] = True
        return self.xyz[point_ids], self.extri[start_idx:end_idx], tracks, masks, point_ids

    def reset_from_tensors(self, points3D, extrinsics, tracks, masks, keep=None, rgb=None, start_idx=0):
        """``reconstruction_to_dicts`` after a joint BA with renumbering (video_runner.py:607-638): the surviving
        points get new dense ids in their old order, observations carry vis = 1."""
        dev = self.device
        P = points3D.shape[0]
        keep = torch.ones(P, dtype=torch.bool, device=dev) if keep is None else keep.to(dev)
        masks = masks.to(dev).bool() & keep[None]
        self.xyz = points3D.to(dev, torch.float64)[keep]
        self.rgb = (torch.zeros((P, 3), device=dev) if rgb is None else rgb.to(dev, torch.float32))[keep]
        new_id = torch.cumsum(keep.long(), 0) - 1
        fr, pi = torch.nonzero(masks, as_tuple=True)
        self.obs_point = torch.zeros(0, dtype=torch.long, device=dev)
        self.obs_frame = torch.zeros(0, dtype=torch.long, device=dev)
        self.obs_uv = torch.zeros((0, 2), dtype=torch.float32, device=dev)
        self.obs_vis = torch.zeros(0, dtype=torch.float32, device=dev)
        self.set_extrinsics(start_idx, extrinsics)
        self._merge_observations(new_id[pi], fr + start_idx, tracks.to(dev)[fr, pi], torch.ones(fr.shape[0], device=dev))
