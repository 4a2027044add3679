"""Device-resident bundle adjustment: host side of the HIP Levenberg-Marquardt solver.

Replaces the round trip ``batch_matrix_to_pycolmap -> pycolmap.bundle_adjustment ->
pycolmap_to_batch_matrix`` of the reference (vggsfm/utils/tensor_to_pycolmap.py:16-214 and the call
sites vggsfm/utils/triangulation.py:197-223,1038-1064,1131-1156): no Python loop over S*P, no host
copies -- the problem is compiled into flat device arrays with torch ops and handed to
``vgg_ba_solve`` (include/vggsfm_amd.h).  Semantics restated from the reference + COLMAP 3.10:

* a 3D point per track with >= 2 masked observations, observations of a point with a coordinate
  >= ``max_points3D_val`` dropped (tensor_to_pycolmap.py:62-146);
* the controller deletes negative-depth observations first; a track of length <= 2 hit by such a
  deletion disappears (its row comes back as zeros, handled at triangulation.py:289-295);
* image 0 pose constant, image 1 x-translation constant; TRIVIAL loss; focal / extra refined,
  principal point constant; one camera per frame or one shared camera.
"""
import copy
import ctypes
from dataclasses import dataclass
from typing import Optional

import math
import os

import numpy as np
import torch

from . import _lib
from .ba_options import LOSS_ID, TERMINATION, BundleAdjustmentOptions

MODEL_ID = {"SIMPLE_PINHOLE": 0, "SIMPLE_RADIAL": 1}
GROUP = 16          # cameras per Schur tile side (must match kGroup in csrc/ba.hip)
CHUNK = 1024        # tile entries per workgroup
SUB = 32            # entries per strided sub-chunk (kSub in csrc/ba.hip)
MIN_CHUNK = 128     # lower bound of the adaptive chunk size
# Single-GPU solves can factorise on a CU-masked stream while the later Schur tile batches run on the other CUs.
# Correct and bit-reproducible, but measured SLOWER on MI355X in round 1 (3.50 vs 3.09 ms per iteration at 200 x 100k:
# the factorisation's trailing updates need 3 rounds on 32 CUs and every dependent launch pays ~7.5 us while a second
# queue is active) -- hence off by default.
OVERLAP_FACTORIZATION = False
OVERLAP_MIN_OBS, OVERLAP_MIN_FRAMES = 1_000_000, 8 * GROUP   # smaller problems keep one batch
TILE_BATCHES = 3     # Schur tile launches per iteration when the factorisation overlaps them (large problems)
CHOL_CUS = 32        # CUs given to the factorisation while it overlaps (options.overlap_factorization; multiple of 32)


# ------------------------------------------------------------------ rotations (Eigen conventions)
def rotmat_to_quat(R):
    """(…,3,3) -> unit quaternion (x,y,z,w), Eigen's Quaternion(Matrix3) branch structure."""
    if R.is_cuda and R.numel() <= 9 * 8192:
        # a few hundred cameras: ~70 tiny launches cost 0.5 ms of every problem set-up; the same formulas on the host
        return _rotmat_to_quat_host(R)
    R = R.to(torch.float64)
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    tr = m00 + m11 + m22
    # trace > 0
    t0 = torch.sqrt(torch.clamp(tr + 1.0, min=1e-300))
    q_tr = torch.stack([(R[..., 2, 1] - R[..., 1, 2]) * (0.5 / t0), (R[..., 0, 2] - R[..., 2, 0]) * (0.5 / t0),
                        (R[..., 1, 0] - R[..., 0, 1]) * (0.5 / t0), 0.5 * t0], -1)
    outs = []
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        t = torch.sqrt(torch.clamp(R[..., i, i] - R[..., j, j] - R[..., k, k] + 1.0, min=1e-300))
        q = [None] * 4
        q[i] = 0.5 * t
        q[3] = (R[..., k, j] - R[..., j, k]) * (0.5 / t)
        q[j] = (R[..., j, i] + R[..., i, j]) * (0.5 / t)
        q[k] = (R[..., k, i] + R[..., i, k]) * (0.5 / t)
        outs.append(torch.stack(q, -1))
    i1 = m11 > m00
    best = torch.where(i1, m11, m00)
    i2 = m22 > best
    q_neg = torch.where(i2[..., None], outs[2], torch.where(i1[..., None], outs[1], outs[0]))
    q = torch.where((tr > 0)[..., None], q_tr, q_neg)
    return q / q.norm(dim=-1, keepdim=True)


def _rotmat_to_quat_host(R):
    import numpy as np
    M = R.detach().to(torch.float64).cpu().numpy()
    m00, m11, m22 = M[..., 0, 0], M[..., 1, 1], M[..., 2, 2]
    tr = m00 + m11 + m22
    t0 = np.sqrt(np.maximum(tr + 1.0, 1e-300))
    q_tr = np.stack([(M[..., 2, 1] - M[..., 1, 2]) * (0.5 / t0), (M[..., 0, 2] - M[..., 2, 0]) * (0.5 / t0),
                     (M[..., 1, 0] - M[..., 0, 1]) * (0.5 / t0), 0.5 * t0], -1)
    outs = []
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(np.maximum(M[..., i, i] - M[..., j, j] - M[..., k, k] + 1.0, 1e-300))
        q = [None] * 4
        q[i] = 0.5 * t
        q[3] = (M[..., k, j] - M[..., j, k]) * (0.5 / t)
        q[j] = (M[..., j, i] + M[..., i, j]) * (0.5 / t)
        q[k] = (M[..., k, i] + M[..., i, k]) * (0.5 / t)
        outs.append(np.stack(q, -1))
    i1 = m11 > m00
    i2 = m22 > np.where(i1, m11, m00)
    q_neg = np.where(i2[..., None], outs[2], np.where(i1[..., None], outs[1], outs[0]))
    q = np.where((tr > 0)[..., None], q_tr, q_neg)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    return torch.from_numpy(np.ascontiguousarray(q)).to(R.device)


def quat_to_rotmat(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = torch.stack([1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, 1 - (txx + tyy)], -1)
    return R.reshape(q.shape[:-1] + (3, 3))


# ------------------------------------------------------------------ problem on the device
@dataclass
class DeviceProblem:
    """Flat device arrays of one BA problem (layout documented in include/vggsfm_amd.h)."""
    cam_q: torch.Tensor
    cam_t: torch.Tensor
    intr: torch.Tensor
    pts: torch.Tensor
    row_ptr: torch.Tensor
    obs_cam: torch.Tensor
    obs_uv: torch.Tensor
    col_ptr: torch.Tensor
    cobs_pt: torch.Tensor
    cobs_uv: torch.Tensor
    chunk_desc: torch.Tensor
    entries: torch.Tensor
    tile_desc: torch.Tensor
    obs_slot: torch.Tensor
    num_segments: int
    camera_model: int
    cam_const: Optional[torch.Tensor] = None
    intr_const: Optional[torch.Tensor] = None
    pt_const: Optional[torch.Tensor] = None
    refine_focal: bool = True
    refine_extra: bool = True
    loss: int = 0
    loss_scale: float = 1.0
    batch_desc: Optional[torch.Tensor] = None   # (B,6) int32 on the HOST (build_schur_tiles); None = derive one batch
    chol_split: tuple = (0, 0)        # (columns of A, columns of B): block-diagonal leading part of the reduced system
    cam_perm: Optional[torch.Tensor] = None     # (S,) long: camera s of this problem is input frame cam_perm[s] (None = identity)
    chol_first_blk: Optional[torch.Tensor] = None   # (ceil(n / 64),) int32 device: row envelope of the reduced system (None = dense)
    merged_tile_launch: bool = False                # off-diagonal and diagonal tile chunks in one launch (small problems)

    @property
    def num_obs(self):
        return int(self.obs_cam.shape[0])

    def c_struct(self):
        P = _lib.BAProblem()
        P.num_cams, P.num_pts, P.num_obs, P.num_intr = (self.cam_t.shape[0], self.pts.shape[0], self.num_obs,
                                                        self.intr.shape[0])
        P.camera_model, P.refine_focal, P.refine_extra = self.camera_model, int(self.refine_focal), int(self.refine_extra)
        P.loss, P.loss_scale = self.loss, self.loss_scale
        for name in ("cam_q", "cam_t", "intr", "pts", "row_ptr", "obs_cam", "obs_uv", "col_ptr", "cobs_pt", "cobs_uv",
                     "cam_const", "intr_const", "pt_const", "chunk_desc", "entries", "tile_desc", "obs_slot"):
            t = getattr(self, name)
            setattr(P, name, None if t is None else t.data_ptr())
        P.num_chunks = self.chunk_desc.shape[0]
        P.merged_tile_launch = int(self.merged_tile_launch)
        if self.batch_desc is None:                 # a hand-built problem: one batch, off-diagonal chunks first
            noff = int((self.chunk_desc[:, 0] != self.chunk_desc[:, 1]).sum().item())
            self.batch_desc = torch.tensor([[0, noff, self.chunk_desc.shape[0], 0, self.tile_desc.shape[0], 0]],
                                           dtype=torch.int32)
        self.batch_desc = self.batch_desc.contiguous()
        P.num_tile_batches = self.batch_desc.shape[0]
        P.tile_batches = self.batch_desc.data_ptr()   # host memory, kept alive by self
        P.num_tiles = self.tile_desc.shape[0]
        P.num_segments = self.num_segments
        P.chol_split_a, P.chol_split_b = int(self.chol_split[0]), int(self.chol_split[1])
        P.chol_first_blk = None if self.chol_first_blk is None else self.chol_first_blk.data_ptr()
        return P


# Measurement hooks.  The product path reads NO environment variable unless VGGSFM_AMD_DEBUG_HOOKS=1 is set (the A/B harness
# scripts/prof/ab_c3.py sets it): then VGGSFM_QUAD_SORT_WINDOW, VGGSFM_TILE_TOP_UP, VGGSFM_TILE_ORDER, VGGSFM_TILE_COST_MODE,
# VGGSFM_TILE_FIXED_COST, VGGSFM_TILE_POS_WEIGHT, VGGSFM_SORT_POINTS, VGGSFM_TILE_WGS and VGGSFM_TILE_BACKFILL override the
# constants below, all read per call (A/B runs switch them in-process).  Every one of them changes the SCHEDULE of the work,
# never a sum: results are bit-identical across their values (tests/test_gpu_ba.py).
def _hook(name, default=None):
    if os.environ.get("VGGSFM_AMD_DEBUG_HOOKS") != "1":
        return default
    return os.environ.get(name, default)


QUAD_SORT_WINDOW = 512      # entries; 0 = plain sweep order inside a tile
# (off-diagonal, diagonal) launch: cost of a tile batch besides its matrix instructions, in matrix instructions of one wavefront
# (staging, LDS write phase, barrier: ~2400 of ~5000 cycles per batch in the round-3 phase trace).  Round 5, same-box sweep on
# configs[2] (profiles/r05_ab_tile_fixed_cost_c3.jsonl): the off-diagonal launch with 6 x 6 blocks is 2.5 % shorter with a
# SMALL fixed part (0 .. 10: 0.543 ms against 0.558 at 18 -- dense tiles get more of the workgroups), the diagonal launch does
# not care between 18 and 200 and loses 15 % at 6.  8 x 8 blocks (one configs[3] shard / configs[3] whole, profiles/
# r05_ab_tile_knobs_c4.jsonl): 8 instead of 18 -3.3 % / -1.2 % on the off-diagonal launch, with the sparsest-first order -6 % / -2.8 %.
TILE_FIXED_COST = {6: (8.0, 18.0), 7: (8.0, 18.0), 8: (8.0, 18.0)}
TILE_TOP_UP = True           # hand the slots the chunk-size search leaves empty to the most loaded tiles
TILE_ORDER = "sparse_first"        # launch order of the tiles: "plain" = (gI, gJ); "dense_first" / "sparse_first" (build_schur_tiles)
TILE_ORDERS = ("plain", "dense_first", "sparse_first")
TILE_ORDERS_EXPERIMENTAL = ("stride", "sdm", "msd", "dsm")     # round-5 A/B variants (profiles/r05_ab_tile_order_c3.jsonl), hooks only
TILE_POSITION_WEIGHT = 0.0        # see build_schur_tiles: extra cost of a tile per unit of launch position (0 = off: with every slot filled
#                                   the sweep 0 .. 0.45 stayed inside the run-to-run noise, profiles/r05_ab_tile_fixed_cost_c3.jsonl)


def _entry_cost(qmask, diag, bd, group=GROUP):
    """Cost model of one tile entry for the workgroup counts (what schur_tile_kernel spends on the QUAD the entry belongs
    to): TILE_FIXED_COST + 3 x the matrix instructions of the busiest of the four wavefronts, from the quad's presence masks
    exactly as the kernel skips -- 16-row block b of a side runs when one of the cameras with rows in it (slots 16 b // bd
    .. (16 b + 15) // bd) is present.  Off-diagonal tile: wavefront (wr, wc) owns the sub-tiles (wr + 2 i, wc + 2 j); diagonal
    tile: the lower-triangle sub-tiles row by row, dealt round-robin.  qmask (E,) long: maskA | maskB << 16; diag (E,) bool."""
    dev = qmask.device
    nt = bd                                                   # 16 bd rows / 16
    bits = []
    for b in range(nt):
        s0, s1 = (16 * b) // bd, min(group - 1, (16 * b + 15) // bd)
        bits.append(((2 << s1) - 1) & ~((1 << s0) - 1))
    bits = torch.tensor(bits, dtype=torch.long, device=dev)
    ra = ((qmask[:, None] & 0xFFFF) & bits[None]) != 0        # (E, nt) row blocks with a camera
    rb = ((qmask[:, None] >> 16) & bits[None]) != 0
    off = torch.maximum(ra[:, 0::2].sum(1), ra[:, 1::2].sum(1)) * torch.maximum(rb[:, 0::2].sum(1), rb[:, 1::2].sum(1))
    if _hook("VGGSFM_TILE_COST_MODE") == "mean":                   # measurement hook: the MEAN wavefront instead of the busiest
        off = (ra.sum(1) * rb.sum(1)).double() / 4.0
    per_wave = torch.zeros((qmask.shape[0], 4), dtype=torch.long, device=dev)
    t = 0
    for r in range(nt):
        for c in range(r + 1):
            per_wave[:, t % 4] += (ra[:, r] & ra[:, c]).long()
            t += 1
    fx = _hook("VGGSFM_TILE_FIXED_COST")                          # measurement hook "off,diag"
    f_off, f_diag = (float(x) for x in fx.split(",")) if fx else TILE_FIXED_COST[bd]
    fixed = torch.where(diag, torch.full_like(off, f_diag, dtype=torch.float64), torch.full_like(off, f_off, dtype=torch.float64))
    return fixed + 3.0 * torch.where(diag, per_wave.max(1).values, off).double()


def build_schur_tiles(row_ptr, obs_cam, group=GROUP, chunk=CHUNK, max_chunks=None, num_batches=1, later_scale=1.0,
                      merged_slots=None, block_rows=6, simple=False, num_cams=None):
    """Block-sparse Schur work list (device, torch ops; structure is fixed for the whole solve).

    A *segment* is the run of one point's observations that falls into one group of `group`
    consecutive cameras; an *entry* pairs two segments (gI <= gJ) of the same point; entries are
    sorted by tile (gI, gJ) and, inside a tile, by the sweep position of the point (first camera, last camera); a tile with k entries gets J = ceil(k / `chunk`)
    workgroups ("chunks", off-diagonal tiles first), workgroup j taking the sub-chunks j, j+J, ... of SUB entries (kSub in ba.hip).
    Returns (chunk_desc (n,6) int32 = gI,gJ,tile_begin,tile_end,j,J ; entries (E,4) int32 = point index,
    segA, segB, maskA|maskB<<16 of the QUAD (the union over the four consecutive entries, aligned to the tile's begin, that one
    MFMA K-step packs; bit l of a mask <=> camera group*16 + l observes one of the four points) ;
    tile_desc (T,4) int32 = gI,gJ,chunk_begin,chunk_end ; obs_slot (O,) int32 = segment*16 + camera%16 ;
    number of segments ; batch_desc (B,6) int32 ON THE HOST).

    Batches: the tiles are cut into `num_batches` runs of consecutive camera groups gI with about equal entry
    counts (batch-major order: inside a batch the off-diagonal tiles, then the diagonal ones).  Column block g of
    the reduced system only receives sums from tiles with gI <= g, so it is complete once the batches up to the one
    of group g are done -- which lets the factorisation start on the first batch while the later ones are still
    being computed.  batch_desc row = (chunk_begin, first_diagonal_chunk, chunk_end, tile_begin, tile_end,
    first_camera_group).  `later_scale` shrinks the workgroup caps of the batches after the first (they run on a
    CU-masked stream beside the factorisation).

    Workgroups per tile: proportional to the tile's COST (_entry_cost: staging + the matrix instructions its presence masks
    leave), not to its entry count -- tiles whose points see few of the 2 x 16 cameras skip most of their sub-tiles and ran
    15 % ahead of the dense ones, the launch waiting for the slowest (round 3 phase trace); `block_rows` = rows per camera
    of the tile blocks (6, or 6 + refined intrinsics when they are per camera).

    `merged_slots`: the off-diagonal and the diagonal chunks will run in ONE launch (vgg_ba_problem.merged_tile_launch) with
    that many resident workgroups in all; the split between the two kinds follows their entry counts.

    `simple` (round 6; compile_problem sets it below SIMPLE_WORKLIST_MAX_OBS observations -- video windows): the launch of a
    small problem takes ~30 us whatever its schedule, while this function is ~400 torch launches and ~35 host reads; the simple
    list keeps the (gI, gJ) tile order and the plain sweep order inside a tile, hands out workgroups by entry count and reads
    the per-tile tables back in one copy each.  Same tiles, same entries per tile, same sums up to their order.
    `num_cams`: the number of cameras (saves the two host reads that derive it from `obs_cam`)."""
    dev = obs_cam.device
    O = obs_cam.shape[0]
    P = row_ptr.shape[0] - 1
    if O == 0:
        z = torch.zeros((0, 4), dtype=torch.int32, device=dev)
        return (torch.zeros((0, 6), dtype=torch.int32, device=dev), z, z.clone(), torch.zeros(0, dtype=torch.int32, device=dev), 0,
                torch.zeros((1, 6), dtype=torch.int32))
    counts = (row_ptr[1:] - row_ptr[:-1]).long()
    obs_pt = torch.repeat_interleave(torch.arange(P, device=dev), counts, output_size=O)   # (output_size: no host read inside)
    grp = (obs_cam // group).long()
    is_start = torch.ones(O, dtype=torch.bool, device=dev)
    is_start[1:] = (obs_pt[1:] != obs_pt[:-1]) | (grp[1:] != grp[:-1])
    seg_begin = torch.nonzero(is_start).squeeze(1)
    nseg = seg_begin.shape[0]
    seg_id = torch.cumsum(is_start.long(), 0) - 1
    # presence mask of the segment: bit l <=> camera group*16 + l observes the point
    seg_mask = torch.zeros(nseg, dtype=torch.long, device=dev).index_add_(0, seg_id, 1 << (obs_cam.long() % group))
    seg_pt = obs_pt[seg_begin]
    seg_grp = grp[seg_begin]
    ncam = int(num_cams) if num_cams is not None else int(obs_cam.max().item()) + 1
    ngroups = (ncam - 1) // group + 1
    pseg_ptr = torch.zeros(P + 1, dtype=torch.long, device=dev)
    pseg_ptr[1:] = torch.cumsum(torch.bincount(seg_pt, minlength=P), 0)
    idx = torch.arange(nseg, device=dev)
    npair = pseg_ptr[seg_pt + 1] - idx
    total = int(npair.sum().item())
    A = torch.repeat_interleave(idx, npair, output_size=total)
    pair_start = torch.cumsum(npair, 0) - npair
    B = A + (torch.arange(total, device=dev) - pair_start[A])
    # batch of a camera group: runs of consecutive groups gI with about equal numbers of entries
    gA, gB = seg_grp[A], seg_grp[B]
    # sweep position of a point: all tiles walk the points in the order (first camera, last camera), so that the four
    # entries a workgroup packs along K (a "quad") have about the same presence pattern (the tile kernels skip the 16-row
    # blocks in which none of the four has a camera) and every tile still meets a given point at about the same time
    has = counts > 0
    first_cam = torch.where(has, obs_cam.long()[row_ptr[:-1].long().clamp(max=O - 1)], torch.zeros_like(counts))
    last_cam = torch.where(has, obs_cam.long()[(row_ptr[1:].long() - 1).clamp(min=0)], torch.zeros_like(counts))
    prank = torch.empty(P, dtype=torch.long, device=dev)
    prank[torch.argsort(first_cam * ncam + last_cam, stable=True)] = torch.arange(P, device=dev)
    nb = max(1, min(int(num_batches), ngroups))
    if nb == 1:                                                     # (one batch: no counts, no host read)
        batch_of_group = torch.zeros(ngroups, dtype=torch.long, device=dev)
    else:
        per_group = torch.bincount(gA, minlength=ngroups).double()
        cum = torch.cumsum(per_group, 0) - per_group                # entries before group g
        batch_of_group = torch.clamp((cum * nb / max(float(total), 1.0)).long(), max=nb - 1)   # (per_group.sum() = total)
        batch_of_group = torch.cummax(batch_of_group, 0).values
    # tile key: batch-major; inside a batch the off-diagonal tiles first, then the diagonal ones (separate launches)
    nn = ngroups * ngroups
    key = batch_of_group[gA] * (2 * nn) + (gA == gB).long() * nn + gA * ngroups + gB
    epos = prank[seg_pt[A]]
    order = torch.argsort(key * P + epos)
    A, B, key, epos = A[order], B[order], key[order], epos[order]
    emask = seg_mask[A] | (seg_mask[B] << 16)
    ukeys, kcounts = torch.unique_consecutive(key, return_counts=True)              # chunking unit = tile
    tile_start = torch.cumsum(kcounts, 0) - kcounts
    order_mode = "plain" if simple else _hook("VGGSFM_TILE_ORDER", TILE_ORDER)
    if order_mode not in TILE_ORDERS + TILE_ORDERS_EXPERIMENTAL:
        raise ValueError(f"tile order {order_mode!r}: expected one of {TILE_ORDERS + TILE_ORDERS_EXPERIMENTAL}")
    if order_mode != "plain" and nb == 1 and ukeys.shape[0] > 2:
        # LAUNCH ORDER of the tiles by density (round 5): the workgroups of a launch are resident three (four) to a CU --
        # positions p, p + CUs, p + 2 CUs -- and the oldest is served first; in (gI, gJ) order neighbouring positions hold
        # tiles of like density.  Ordered by the mean number of 16-row block products of their entries (own patterns), the
        # launch walks from the sparsest tiles to the densest and every CU gets a mix.  Same-box A/B at configs[2]
        # (profiles/r05_ab_tile_order_c3.jsonl): densest first 0.564 / 0.247 ms (off-diagonal / diagonal launch), (gI, gJ)
        # order 0.542 / 0.252, SPARSEST FIRST 0.536 / 0.243 -- the default; densities scattered over the launch 0.570 /
        # 0.244.  The sums of a tile do not depend on where it is launched (final costs equal to the last bit).
        # (per-tile sums of a list sorted by tile = differences of a running sum; the runs are then moved as blocks -- no
        #  second sort and no atomics: an index_add_ of 6 M doubles into ~100 bins takes seconds)
        nt_ = block_rows * group // 16
        bits_ = torch.tensor([(((2 << min(group - 1, (16 * b + 15) // block_rows)) - 1) & ~((1 << ((16 * b) // block_rows)) - 1))
                              for b in range(nt_)], dtype=torch.long, device=dev)
        nblk = lambda m: ((m[:, None] & bits_[None]) != 0).sum(1)
        csum_d = torch.cumsum((nblk(seg_mask[A]) * nblk(seg_mask[B])).double(), 0)
        tend_d = tile_start + kcounts - 1
        dsum = csum_d[tend_d] - torch.where(tile_start > 0, csum_d[(tile_start - 1).clamp(min=0)], torch.zeros_like(csum_d[tend_d]))
        dens = dsum / kcounts.double()
        cls = ukeys // nn                                              # (batch, diagonal flag): the launches keep their order
        span = float(dens.max().item()) + 1.0
        skey = cls.double() * (2.0 * span) + (span - dens if order_mode == "dense_first" else dens)
        perm_units = torch.argsort(skey, stable=True)
        if order_mode == "stride":                                       # (experiment: densities scattered over a launch)
            ncls = torch.bincount(cls)
            cstart = torch.cumsum(ncls, 0) - ncls
            r = torch.arange(perm_units.shape[0], device=dev) - cstart[cls[perm_units]]
            nc = ncls[cls[perm_units]]
            step = torch.clamp((nc.double().sqrt().long() | 1), min=1)
            while bool((torch.gcd(step, nc) != 1).any()):
                step = torch.where(torch.gcd(step, nc) != 1, step + 2, step)
            newr = (r * step) % nc
            perm_units = perm_units[torch.argsort(cstart[cls[perm_units]] + newr, stable=True)]
        if order_mode in ("sdm", "msd", "dsm"):                          # (experiment: the density thirds of a launch in another order)
            ncls = torch.bincount(cls)
            cstart = torch.cumsum(ncls, 0) - ncls
            cp = cls[perm_units]
            r = torch.arange(perm_units.shape[0], device=dev) - cstart[cp]
            third = torch.clamp((3 * r) // ncls[cp].clamp(min=1), max=2)       # 0 sparse, 1 mid, 2 dense
            slot = {"sdm": (0, 2, 1), "msd": (1, 0, 2), "dsm": (2, 0, 1)}[order_mode]
            where = torch.tensor([slot.index(t) for t in range(3)], device=dev)[third]
            perm_units = perm_units[torch.argsort(cp * 4 + where, stable=True)]
        new_start = torch.empty_like(tile_start)
        kc_p = kcounts[perm_units]
        new_start[perm_units] = torch.cumsum(kc_p, 0) - kc_p
        unit0 = torch.repeat_interleave(torch.arange(kcounts.shape[0], device=dev), kcounts, output_size=total)
        dest = new_start[unit0] + (torch.arange(total, device=dev) - tile_start[unit0])
        inv = torch.empty_like(dest)
        inv[dest] = torch.arange(total, device=dev)
        A, B, key, epos, emask = A[inv], B[inv], key[inv], epos[inv], emask[inv]
        ukeys, kcounts = ukeys[perm_units], kc_p
        tile_start = torch.cumsum(kcounts, 0) - kcounts
    quad_window = 0 if simple else int(_hook("VGGSFM_QUAD_SORT_WINDOW", QUAD_SORT_WINDOW))
    top_up = (not simple) and _hook("VGGSFM_TILE_TOP_UP", "1" if TILE_TOP_UP else "0") != "0"
    if quad_window > 0:
        # Inside windows of QUAD_SORT_WINDOW consecutive entries of a tile, entries with the same pattern of 16-row blocks
        # (what the tile kernel can skip) are put next to each other: a quad's union is then the pattern of each of its four
        # entries.  Points with the same first camera end in different 16-row blocks of the last group and vice versa, so in
        # plain sweep order a quad ran 5.2 of 6 blocks per side where its entries own 4.85 (c3: 12 % of the matrix instructions).
        nt = block_rows * group // 16
        bits = [(((2 << min(group - 1, (16 * b + 15) // block_rows)) - 1) & ~((1 << ((16 * b) // block_rows)) - 1)) for b in range(nt)]
        bits_t = torch.tensor(bits, dtype=torch.long, device=dev)
        pat = lambda m: (((m[:, None] & bits_t[None]) != 0).long() << torch.arange(nt, device=dev)[None]).sum(1)
        pkey = pat(seg_mask[A]) * (1 << nt) + pat(seg_mask[B])
        unit0 = torch.repeat_interleave(torch.arange(kcounts.shape[0], device=dev), kcounts, output_size=total)
        upos0 = torch.arange(total, device=dev) - tile_start[unit0]
        wkey = (unit0 * (int(kcounts.max().item()) // quad_window + 1) + upos0 // quad_window) * (1 << (2 * nt)) + pkey
        order2 = torch.argsort(wkey, stable=True)
        A, B, epos, emask = A[order2], B[order2], epos[order2], emask[order2]
    # presence of a quad = union over its four entries (quads are aligned to the start of the unit, like the kernel's batches)
    unit_of_entry = torch.repeat_interleave(torch.arange(kcounts.shape[0], device=dev), kcounts, output_size=total)
    upos = torch.arange(total, device=dev) - tile_start[unit_of_entry]
    nquad = (kcounts + 3) // 4
    quad = (torch.cumsum(nquad, 0) - nquad)[unit_of_entry] + upos // 4
    # (rows: an upper bound of the quad count that is known on the host -- sum ceil(k / 4) <= (total + 3 units) / 4 -- instead of
    #  reading the sum back; the rows beyond the last quad are never gathered)
    qm = torch.zeros(((total + 3 * int(kcounts.shape[0])) // 4 + 1, 4), dtype=torch.long, device=dev)
    qm[quad, upos % 4] = emask
    qm = qm[:, 0] | qm[:, 1] | qm[:, 2] | qm[:, 3]
    # (field 0 = the entry's POINT, row of `pts`: the diagonal tile launch fetches the point's back-substitution block by it --
    #  vgg_ba_set_tile_rhs; the sweep position only orders the list)
    entries = torch.stack([seg_pt[A], A, B, qm[quad]], 1).to(torch.int32)
    tbatch = ukeys // (2 * nn)
    is_diag = (ukeys % (2 * nn)) >= nn
    ukeys = ukeys % nn
    if simple:
        kw = kcounts.double()                               # workgroups by entry count
    else:
        # cost-weighted entry count of every tile (relative to the mean entry of its launch kind)
        ecost = _entry_cost(qm[quad], is_diag[unit_of_entry], block_rows, group)
        # (per-tile sums of a list that is sorted by tile: differences of a running sum -- index_add_ with 5 M double atomics
        #  cost 2.4 ms of a 10 ms compile at configs[2])
        csum = torch.cumsum(ecost, 0)
        tend = tile_start + kcounts - 1
        tcost = csum[tend] - torch.where(tile_start > 0, csum[(tile_start - 1).clamp(min=0)], torch.zeros_like(csum[tend]))
        kweight = torch.ones_like(tcost)
        for sel in (is_diag, ~is_diag):
            if bool(sel.any()):
                kweight[sel] = (tcost[sel] / kcounts[sel].double()) / (tcost[sel].sum() / kcounts[sel].sum().double())
        kw = kcounts.double() * kweight
    # (upper bound of a tile's workgroups: at least MIN_CHUNK entries each -- one sub-chunk at the very least)
    nsub_t = torch.clamp(torch.minimum((kcounts + SUB - 1) // SUB, (kcounts + MIN_CHUNK - 1) // MIN_CHUNK), min=1)

    def size_chunks(kw):
        """workgroups per tile for the cost-weighted entry counts kw"""
        csize = torch.full_like(kcounts, chunk)
        if max_chunks is not None:
            # one resident round per launch: the smallest chunk size (multiple of SUB, >= MIN_CHUNK) whose workgroup
            # count fits the device (CUs x workgroups per CU), so no workgroup starts late and runs alone on its CU
            caps = max_chunks if isinstance(max_chunks, (tuple, list)) else (max_chunks, max_chunks)
            if merged_slots is not None:
                # ONE launch for both kinds of tiles (small problems): its resident slots are shared in proportion to the
                # staged bytes (two segments per off-diagonal entry, one per diagonal entry)
                kc0, dg0 = ht[0].astype(np.int64), ht[1] != 0
                e_off, e_diag = float(kc0[~dg0].sum()), float(kc0[dg0].sum())
                n_off = int(round(merged_slots * 2.0 * e_off / max(2.0 * e_off + e_diag, 1.0)))
                n_off = min(max(n_off, 1 if e_off else 0), merged_slots - (1 if e_diag else 0))
                caps = (max(n_off, 1), max(merged_slots - n_off, 1))
            # (the search runs on host copies of the per-tile counts: one synchronisation instead of one per probe -- and, since
            #  round 6, in numpy: a dozen probes of a handful of tiny torch CPU tensors cost 0.4 ms of a 2.9 ms window compile)
            kc_h, diag_h, tb_h, ns_h = ht[0].astype(np.int64), ht[1] != 0, ht[2].astype(np.int64), ht[3].astype(np.int64)
            kw_h = kw.cpu().numpy() if kw is not kw_first else ht[4]
            csize_h = np.full_like(kc_h, chunk)
            topup_h = np.zeros_like(kc_h)
            for b in range(nb):
                scale = 1.0 if b == 0 else later_scale
                for sel, cap in ((~diag_h & (tb_h == b), int(caps[0] * scale)), (diag_h & (tb_h == b), int(caps[1] * scale))):
                    if not bool(sel.any()):
                        continue
                    kc, kwv, nsv = kc_h[sel], kw_h[sel], ns_h[sel]
                    lo, hi = MIN_CHUNK // SUB, max(MIN_CHUNK // SUB, int(-(-int(max(float(kwv.max()), float(kc.max()))) // SUB)))
                    fits = lambda c: int(np.minimum(np.maximum(np.ceil(kwv / (c * SUB)), 1.0).astype(np.int64), nsv).sum()) <= cap
                    if not fits(hi):
                        lo = hi                         # more tiles than slots: one workgroup per tile
                    while lo < hi:
                        mid = (lo + hi) // 2
                        lo, hi = (lo, mid) if fits(mid) else (mid + 1, hi)
                    csize_h[sel] = lo * SUB
                    # top-up (round 5): the search stops at the first chunk size that fits, which leaves 1-3 % of the slots of
                    # a launch empty; they go, one by one, to the tile whose workgroups carry the most
                    n_h = np.minimum(np.maximum(np.ceil(kwv / (lo * SUB)), 1.0).astype(np.int64), nsv)
                    spare = cap - int(n_h.sum())
                    if top_up and 0 < spare <= 64:
                        for _ in range(spare):
                            load = np.where(n_h < nsv, kwv / n_h.astype(np.float64), 0.0)
                            i = int(np.argmax(load))
                            if float(load[i]) <= 0.0:
                                break
                            n_h[i] += 1
                        topup_h[sel] = n_h
            csize = torch.from_numpy(csize_h).to(dev)
        nchunks = torch.minimum(torch.clamp(torch.ceil(kw / csize.double()), min=1).long(), nsub_t)
        if max_chunks is not None and bool((topup_h > 0).any()):
            th = torch.from_numpy(topup_h).to(dev)
            nchunks = torch.where(th > 0, th, nchunks)
        return nchunks

    # (the per-tile tables the search reads: ONE copy to the host -- they are exact in float64)
    kw_first = kw
    host_tables = torch.stack([kcounts.double(), is_diag.double(), tbatch.double(), nsub_t.double(), kw]).cpu()
    ht = host_tables.numpy()
    nchunks = size_chunks(kw)
    # POSITION weight (round 5): the workgroups of a launch are all resident at once, three (four) to a CU, and the SIMDs
    # arbitrate oldest-first -- the workgroup that was dispatched LAST onto a CU gets the issue slots its elders leave.  Per-tile
    # phase trace at configs[2] (scripts/prof/tile_cost_fit.py): 4450 cycles per batch in the first third of the launch, 4800
    # in the second, 5180 in the third, for the same matrix instructions per batch.  So a tile's cost grows with the position
    # of its workgroups in the launch: second pass with kw (1 + TILE_POSITION_WEIGHT x position fraction within its launch).
    pw = 0.0 if simple else float(_hook("VGGSFM_TILE_POS_WEIGHT", TILE_POSITION_WEIGHT))
    if pw != 0.0 and max_chunks is not None and kcounts.shape[0] > 1:
        launch_key = tbatch * 2 + is_diag.long()                   # (units are sorted by batch, off-diagonal first)
        cum = torch.cumsum(nchunks, 0).double()
        total_of = torch.zeros(int(launch_key.max().item()) + 1, dtype=torch.float64, device=dev).index_add_(0, launch_key, nchunks.double())
        starts = torch.cumsum(total_of, 0) - total_of
        frac = ((cum - 0.5 * nchunks.double()) - starts[launch_key]) / total_of[launch_key].clamp(min=1.0)
        nchunks = size_chunks(kw * (1.0 + pw * frac))
    ctile = torch.repeat_interleave(torch.arange(ukeys.shape[0], device=dev), nchunks)          # unit of every chunk
    cfirst = torch.cumsum(nchunks, 0) - nchunks
    local = torch.arange(ctile.shape[0], device=dev) - cfirst[ctile]
    chunk_desc = torch.stack([ukeys[ctile] // ngroups, ukeys[ctile] % ngroups, tile_start[ctile],
                              tile_start[ctile] + kcounts[ctile], local, nchunks[ctile]], 1).to(torch.int32)
    # tiles: runs of consecutive units with the same tile key (the partial tiles of ALL their chunks are summed)
    ukey_full = tbatch * (2 * nn) + is_diag.long() * nn + ukeys
    new_tile = torch.ones_like(ukey_full, dtype=torch.bool)
    new_tile[1:] = ukey_full[1:] != ukey_full[:-1]
    tfirst_unit = torch.nonzero(new_tile).squeeze(1)
    tile_of_unit = torch.cumsum(new_tile.long(), 0) - 1
    t_chunks = torch.zeros(tfirst_unit.shape[0], dtype=torch.long, device=dev).index_add_(0, tile_of_unit, nchunks)
    t_cfirst = cfirst[tfirst_unit]
    tile_desc = torch.stack([ukeys[tfirst_unit] // ngroups, ukeys[tfirst_unit] % ngroups, t_cfirst, t_cfirst + t_chunks],
                            1).to(torch.int32)
    obs_slot = (seg_id * group + obs_cam.long() % group).to(torch.int32)
    # host-side batch table (per tile)
    tbatch, is_diag, ukeys = tbatch[tfirst_unit], is_diag[tfirst_unit], ukeys[tfirst_unit]
    tt = torch.stack([tbatch, is_diag.long(), t_chunks, t_cfirst, ukeys // ngroups]).cpu().numpy()      # (one copy)
    tb, td, tn, tf, tg = tt[0], tt[1] != 0, tt[2], tt[3], tt[4]
    bd = np.zeros((nb, 6), dtype=np.int32)
    for b in range(nb):
        ids = np.nonzero(tb == b)[0]
        if ids.size == 0:                           # (cannot happen for b = 0; later batches may be empty)
            prev = int(bd[b - 1, 2]) if b else 0
            prevt = int(bd[b - 1, 4]) if b else 0
            bd[b] = (prev, prev, prev, prevt, prevt, ngroups)
            continue
        c0 = int(tf[ids[0]])
        c1 = int(tf[ids[-1]] + tn[ids[-1]])
        dg = ids[td[ids]]
        cm = int(tf[dg[0]]) if dg.size else c1
        bd[b] = (c0, cm, c1, int(ids[0]), int(ids[-1]) + 1, int(tg[ids].min()))
    batch_desc = torch.from_numpy(bd)
    return (chunk_desc.contiguous(), entries.contiguous(), tile_desc.contiguous(), obs_slot.contiguous(), int(nseg),
            batch_desc.contiguous())


SORT_POINTS = True                 # bundle_adjustment: number the problem's points by track length (compile_problem); hook VGGSFM_SORT_POINTS=0
TILE_BACKFILL = None               # (off-diagonal, diagonal) workgroups per CU of the single back-filled tile launch; None = two launches
TILE_WGS_PER_CU = (3, 4)           # resident schur_tile workgroups per CU with 6 x 6 blocks: (off-diagonal, diagonal) launch
SPARSE_GRID_DENSITY = 0.05        # compile_problem: below this fill of the (frames x tracks) grid work on the observation list
MERGED_TILE_MAX_OBS = 1_000_000   # below: off-diagonal and diagonal tiles share one launch
SMALL_GRID_CELLS = 1 << 20        # compile_problem: (frames x tracks) cells up to which the grid path is taken without looking at its fill
SIMPLE_WORKLIST_MAX_OBS = 100_000 # below: build_schur_tiles(simple=True) -- no density order, pattern grouping or cost model (video windows)
CAMERA_SPLIT_MIN_STEPS = 2    # shared 64-column factorisation steps below which re-ordering the cameras is not worth it


def find_camera_split(masks, group=GROUP, adjacency_reduce=None):
    """Sliding-window visibility (video, ordered image sets): no point is seen both by the first cameras and by the
    last ones, so the reduced camera system has a block-diagonal leading part once those two sets are ordered first --
    and the factorisations of the two blocks can advance side by side (csrc/chol.hip).  masks (S,P) bool.
    Returns (perm (S,) long new -> input frame | None, (columns of A, columns of B)).  A = the first `a` camera groups
    (a even, so that A's 96 a columns are a multiple of 64), B = every FULL group behind the last one that shares a point
    with A (a partial last group goes to the very end, so that the 16-camera groups of the tile kernels stay aligned);
    the pair with the most shared 64-column steps wins.  `adjacency_reduce(t)`: in-place MAX all-reduce of the
    (G,G) group adjacency -- with several ranks the structure must hold for the SUM of their systems."""
    S = masks.shape[0]
    G = (S + group - 1) // group
    if G < 4:
        return None, (0, 0)
    adj = _group_adjacency(masks, group, adjacency_reduce)
    Gfull = S // group                              # B takes whole groups only: every group of the new order stays aligned
    best = (0, 0, 0)
    for a in range(2, Gfull - 1, 2):
        reach = int(torch.nonzero(adj[:a].any(0)).max())               # last group that shares a point with A
        b = max(reach + 1, a)
        if b >= Gfull:
            break
        steps = min(6 * group * a, 6 * group * (Gfull - b)) // 64
        if steps > best[0]:
            best = (steps, a, b)
    steps, a, b = best
    if steps < CAMERA_SPLIT_MIN_STEPS:
        return None, (0, 0)
    dev = masks.device
    perm = torch.cat([torch.arange(0, group * a, device=dev), torch.arange(group * b, group * Gfull, device=dev),
                      torch.arange(group * a, group * b, device=dev), torch.arange(group * Gfull, S, device=dev)])
    return perm, (6 * group * a, 6 * group * (Gfull - b))


def _group_adjacency(masks, group, adjacency_reduce):
    """(G, G) bool on the host: do camera groups g and h share a point?  Without a matrix product -- the first GEMM of a process
    costs a 160 ms code-object load in the BLAS library (measured inside the second joint BA of the configs[4] loop, round 6),
    more than every other launch of that call together.  The points' group sets are packed into 62-bit words, the DISTINCT sets
    (a few thousand at most for video-like visibility) unpacked again and combined pairwise."""
    S = masks.shape[0]
    G = (S + group - 1) // group
    pad = G * group - S
    dev = masks.device
    m = torch.cat([masks, masks.new_zeros((pad, masks.shape[1]))]) if pad else masks
    V = m.reshape(G, group, -1).any(1)                                     # (G, P) bool
    words = []
    for w0 in range(0, G, 62):
        Vw = V[w0:w0 + 62]
        pow2 = (torch.ones((), dtype=torch.int64, device=dev) << torch.arange(Vw.shape[0], device=dev, dtype=torch.int64))[:, None]
        words.append((Vw.to(torch.int64) * pow2).sum(0))                   # (P,)
    uniq = torch.unique(words[0])[:, None] if len(words) == 1 else torch.unique(torch.stack(words, 1), dim=0)
    bits = [((uniq[:, i:i + 1] >> torch.arange(min(62, G - 62 * i), device=dev, dtype=torch.int64)) & 1).bool()
            for i in range(len(words))]
    Vu = torch.cat(bits, 1)                                                # (U, G) the distinct group sets
    adj = torch.zeros((G, G), dtype=torch.bool, device=dev)
    step = max(1, (1 << 25) // (G * G))                                    # <= 32 MB of pairwise products at a time
    for i in range(0, Vu.shape[0], step):
        blk = Vu[i:i + step]
        adj |= (blk[:, :, None] & blk[:, None, :]).any(0)
    if adjacency_reduce is not None:
        adj_f = adj.to(torch.float32)
        adjacency_reduce(adj_f)
        return adj_f.cpu() > 0
    return adj.cpu()


CAMERA_ORDER_MIN_GAIN = 0.8     # use the k-way order when its pivot chain is at most this fraction of the 2-way one


def find_camera_order(masks, group=GROUP, adjacency_reduce=None):
    """k-way generalisation of :func:`find_camera_split` for banded visibility (video: a track spans a few dozen of a
    thousand frames).  With a band of w camera groups, k interior runs of groups separated by k - 1 separators of w
    groups do not couple with each other; ordered [interior_1 .. interior_k, separator_1 .. separator_{k-1}, partial last
    group] the reduced system has k independent leading blocks, and what is left -- the separators -- is block
    tridiagonal.  The dataflow factorisation needs no special casing for this: it is told the ROW ENVELOPE of the
    permuted matrix (``chol_first_blk``) and factors whatever does not depend on each other concurrently; the pivot
    chain shrinks from all G groups to (interior + all separators).
    Returns (perm (S,) long new -> input frame, first_group (G,) long: for the group at new position p the first new
    position it couples with) or (None, None) when the order would not pay."""
    S = masks.shape[0]
    G = (S + group - 1) // group
    Gfull = S // group
    if Gfull < 6:
        return None, None
    adj = _group_adjacency(masks, group, adjacency_reduce)
    ii, jj = torch.nonzero(adj[:Gfull, :Gfull], as_tuple=True)
    w = int((ii - jj).abs().max()) if ii.numel() else 0          # band half-width in groups
    if w < 1 or 2 * w + 2 > Gfull:
        return None, None
    best = None
    for k in range(2, Gfull):
        sep = (k - 1) * w
        rem = Gfull - sep
        if rem < k:
            break
        chain = -(-rem // k) + sep                                # longest interior + every separator
        if best is None or chain < best[0]:
            best = (chain, k)
    two_way = -(-(Gfull - w) // 2) + w
    if best is None or best[1] <= 2 or best[0] > CAMERA_ORDER_MIN_GAIN * two_way:
        return None, None
    chain, k = best
    rem = Gfull - (k - 1) * w
    # interior runs of an EVEN number of groups (96 columns each): every run then starts on a 64-column block of the
    # factorisation -- a run that starts inside a block shares that block with its neighbour and the two pivot chains
    # become one (c5: four runs, two chains).  An odd group left over goes to the last run.
    pairs = rem // 2
    sizes = [2 * (pairs // k + (1 if j < pairs % k else 0)) for j in range(k)]
    sizes[-1] += rem - 2 * pairs
    if min(sizes) < 1:
        return None, None
    interiors, separators, pos = [], [], 0
    for j in range(k):
        interiors.append(list(range(pos, pos + sizes[j])))
        pos += sizes[j]
        if j < k - 1:
            separators.append(list(range(pos, pos + w)))
            pos += w
    order = [g for run in interiors for g in run] + [g for run in separators for g in run] + list(range(Gfull, G))
    order_t = torch.tensor(order)
    adjp = adj[order_t][:, order_t]
    first_group = torch.tensor([int(torch.nonzero(adjp[p, :p + 1])[0]) if bool(adjp[p, :p + 1].any()) else p for p in range(G)])
    dev = masks.device
    perm = torch.cat([torch.arange(group * g, min(group * (g + 1), S), device=dev) for g in order])
    return perm, first_group


def envelope_blocks(first_group, num_cams, n_reduced, group=GROUP, block=64):
    """Row envelope of the reduced system in `block`-column blocks from the camera-group structure: pose rows of the
    cameras of the group at position p start at column 6 * group * first_group[p]; the intrinsics rows behind the 6 C
    pose columns (and the appended right-hand side) are dense.  -> (ceil(n / block),) int32."""
    nbk = (n_reduced + block - 1) // block
    rows = torch.arange(nbk * block)
    cam = torch.clamp(rows // 6, max=num_cams - 1)
    first_col = torch.where(rows < 6 * num_cams, 6 * group * first_group[cam // group], torch.zeros_like(rows))
    return (first_col.reshape(nbk, block).min(1).values // block).to(torch.int32)


def compile_problem(points3d, extrinsics, intrinsics, tracks, masks, extra_params=None, shared_camera=False,
                    camera_type="SIMPLE_PINHOLE", max_points3D_val=3000, filter_negative_depth=True,
                    gauge="colmap", overlap=None, camera_split=False, adjacency_reduce=None, refine_focal_length=True,
                    refine_extra_params=True, sort_points=False):
    """tensors (reference layout, on the GPU) -> DeviceProblem + bookkeeping.
    Returns (problem, valid_idx (P',) long, deleted (P',) bool).
    overlap: cut the Schur tiles into TILE_BATCHES batches so that a single-GPU solve can factorise beside the later
    ones (default OVERLAP_FACTORIZATION; pass False for a multi-GPU shard, whose system is all-reduced first).
    camera_split: look for a block-diagonal leading part (find_camera_split) and order the cameras accordingly --
    problem.cam_perm then maps the problem's cameras to the input frames (cam_q / cam_t / per-camera intr / cam_const
    are in THAT order).  With several ranks pass `adjacency_reduce` so that every rank orders alike.
    refine_focal_length / refine_extra_params: what the solve will refine (``BundleAdjustmentOptions``) -- they decide the
    rows per camera of the Schur tile blocks (6, or 6 + refined intrinsics when those are per camera), hence the tile
    kernel variant, its occupancy and the balance of the work list (the video window BA refines neither).
    sort_points (round 5): the problem's points are numbered by TRACK LENGTH (ascending, ties in track order) instead of in
    track order -- `valid_idx` is then not monotonic.  The point passes give 16 (32) lanes to a point and 4 (2) points to a
    wavefront, which walks the longest of its tracks: with neighbours of equal length no lane waits for another point's
    observations.  Nothing else depends on the numbering (the tile work list is ordered by sweep position, the camera-major
    list by camera)."""
    if camera_type not in MODEL_ID:
        raise ValueError(f"Camera type {camera_type} is not supported yet")
    dev = tracks.device
    S = extrinsics.shape[0]
    ext = extrinsics.to(torch.float64)
    K = intrinsics.to(torch.float64)
    masks = masks.bool()
    cam_perm, chol_split, first_group = (None, (0, 0), None)
    if camera_split:
        cam_perm, first_group = find_camera_order(masks, adjacency_reduce=adjacency_reduce)       # k > 2 interior runs
        if cam_perm is None:
            cam_perm, chol_split = find_camera_split(masks, adjacency_reduce=adjacency_reduce)   # two leading blocks
    # Two constructions of the same arrays.  A (frames x tracks) grid that is mostly empty -- video: 1000 x 250 k slots for
    # 4.7 M observations, 2 % -- is turned into the LIST of its observations first and everything works on that (a dozen
    # passes over the grid, the depth of every slot alone 2 GB, cost more than the solve they prepare: configs[4] 3.5 -> 3.2 s);
    # a well-filled grid (batch configs: 25 %) is cheaper to filter in place (c3: 5.6 ms against 8 ms for the list).
    # (a small grid is filtered in place whatever its fill: no host read for the decision)
    if masks.numel() > SMALL_GRID_CELLS and float(masks.sum().item()) < SPARSE_GRID_DENSITY * masks.numel():
        f0, n0 = torch.nonzero(masks, as_tuple=True)                   # input frame, input track; frame-major
        if cam_perm is not None:                        # (frames 0 and 1 -- the default gauge -- stay first: A is a prefix)
            ext, K = ext[cam_perm], K[cam_perm]
            if extra_params is not None:
                extra_params = extra_params[cam_perm]
            inv = torch.empty_like(cam_perm)
            inv[cam_perm] = torch.arange(S, device=dev)
            fs = inv[f0]                                               # position of the frame in the problem's camera order
            o = torch.argsort(fs, stable=True)                         # camera-major in THAT order (tracks ascending inside)
            f0, n0, fs = f0[o], n0[o], fs[o]
        else:
            fs = f0
        N = masks.shape[1]
        length0 = torch.bincount(n0, minlength=N)
        valid = length0 >= 2
        valid_idx = torch.nonzero(valid).squeeze(1)
        if sort_points:
            valid_idx = valid_idx[torch.argsort(length0[valid_idx], stable=True)]
            newid = torch.full((N,), -1, dtype=torch.long, device=dev)
            newid[valid_idx] = torch.arange(valid_idx.shape[0], device=dev)
        else:
            newid = torch.cumsum(valid.long(), 0) - 1
        pts = points3d.to(torch.float64)[valid_idx].contiguous()
        P = pts.shape[0]
        ok = valid[n0]
        f0, fs, p = f0[ok], fs[ok], newid[n0[ok]]
        ok = (pts < max_points3D_val).all(-1)[p]                       # observations of points with a coordinate >= the cap do not enter
        f0, fs, p = f0[ok], fs[ok], p[ok]
        deleted = torch.zeros(P, dtype=torch.bool, device=dev)
        if filter_negative_depth:
            # ObservationManager::FilterObservationsWithNegativeDepth, vectorised: observations are visited
            # image by image; one is deleted when its depth < eps, and the whole point goes once a deletion
            # meets a track of length <= 2  <=>  (initial length - number of bad observations) <= 1.
            z = (ext[fs, 2, :3] * pts[p]).sum(-1) + ext[fs, 2, 3]
            bad = ~(z >= torch.finfo(torch.float64).eps)
            nbad = torch.bincount(p[bad], minlength=P)
            deleted = (nbad >= 1) & ((torch.bincount(p, minlength=P) - nbad) <= 1)
            ok = ~bad & ~deleted[p]
            f0, fs, p = f0[ok], fs[ok], p[ok]
        tr = tracks.to(torch.float32)
        # camera-major: sorted by frame (problem order); inside a frame in the order of the list -- by input track, which is
        # by point only without sort_points (no consumer needs the points of a camera ascending: the camera passes sum a
        # camera's observations in list order, whatever it is)
        cobs_pt = p.to(torch.int32).contiguous()
        col_ptr = torch.zeros(S + 1, dtype=torch.int32, device=dev)
        col_ptr[1:] = torch.cumsum(torch.bincount(fs, minlength=S), 0).to(torch.int32)
        cobs_uv = tr[f0, valid_idx[p]].contiguous()
        # point-major: sorted by point, then frame
        o = torch.argsort(p, stable=True)
        obs_cam = fs[o].to(torch.int32).contiguous()
        row_ptr = torch.zeros(P + 1, dtype=torch.int32, device=dev)
        row_ptr[1:] = torch.cumsum(torch.bincount(p, minlength=P), 0).to(torch.int32)
        obs_uv = cobs_uv[o].contiguous()
    else:
        if cam_perm is not None:                        # (frames 0 and 1 -- the default gauge -- stay first: A is a prefix)
            ext, K, masks, tracks = ext[cam_perm], K[cam_perm], masks[cam_perm], tracks[cam_perm]
            if extra_params is not None:
                extra_params = extra_params[cam_perm]
        length0 = masks.sum(0)
        valid_idx = torch.nonzero(length0 >= 2).squeeze(1)
        if sort_points:
            valid_idx = valid_idx[torch.argsort(length0[valid_idx], stable=True)]
        pts = points3d.to(torch.float64)[valid_idx].contiguous()
        m = masks[:, valid_idx].clone()
        m[:, ~(pts < max_points3D_val).all(-1)] = False
        deleted = torch.zeros(valid_idx.shape[0], dtype=torch.bool, device=dev)
        if filter_negative_depth:
            # ObservationManager::FilterObservationsWithNegativeDepth, vectorised: observations are visited
            # image by image; one is deleted when its depth < eps, and the whole point goes once a deletion
            # meets a track of length <= 2  <=>  (initial length - number of bad observations) <= 1.
            z = torch.einsum("sj,pj->sp", ext[:, 2, :3], pts) + ext[:, 2, 3][:, None]
            bad = m & ~(z >= torch.finfo(torch.float64).eps)
            nbad = bad.sum(0)
            deleted = (nbad >= 1) & ((m.sum(0) - nbad) <= 1)
            m = m & ~bad
            m[:, deleted] = False
        P = pts.shape[0]
        pm = torch.nonzero(m.t())                       # point-major: sorted by point, then frame
        obs_cam = pm[:, 1].to(torch.int32).contiguous()
        counts = m.sum(0)
        row_ptr = torch.zeros(P + 1, dtype=torch.int32, device=dev)
        row_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        tr = tracks.to(torch.float32)
        obs_uv = tr[pm[:, 1], valid_idx[pm[:, 0]]].contiguous()
        cm = torch.nonzero(m)                           # camera-major: sorted by frame, then point
        cobs_pt = cm[:, 1].to(torch.int32).contiguous()
        col_ptr = torch.zeros(S + 1, dtype=torch.int32, device=dev)
        col_ptr[1:] = torch.cumsum(m.sum(1), 0).to(torch.int32)
        cobs_uv = tr[cm[:, 0], valid_idx[cm[:, 1]]].contiguous()
    cam_q = rotmat_to_quat(ext[:, :, :3]).contiguous()
    cam_t = ext[:, :, 3].contiguous()
    n_intr = 1 if shared_camera else S
    intr = torch.zeros((n_intr, 4), dtype=torch.float64, device=dev)
    src = slice(0, 1) if shared_camera else slice(None)
    intr[:, 0] = K[src, 0, 0]
    intr[:, 1] = K[src, 0, 2]
    intr[:, 2] = K[src, 1, 2]
    if camera_type == "SIMPLE_RADIAL":
        intr[:, 3] = extra_params.to(torch.float64)[src, 0]
    cam_const = torch.zeros(S, dtype=torch.uint8, device=dev)
    if gauge == "colmap":
        cam_const[0] = 1                            # SetConstantCamPose(first registered image)
        if S > 1:
            cam_const[1] = 2                        # SetConstantCamPositions(second image, {0})
    cus = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 256
    # resident schur_tile workgroups per CU (occupancy of the kernel variant): off-diagonal launch 3 (BD = 6) or 2,
    # diagonal launch 4 or 2 -- one full round each
    kd = int(bool(refine_focal_length)) + int(bool(refine_extra_params) and camera_type == "SIMPLE_RADIAL")
    block_rows = 6 if (shared_camera or kd == 0) else 6 + kd          # BD of schur_tile_kernel (make_dims in csrc/ba.hip)
    slots = (cus * TILE_WGS_PER_CU[0], cus * TILE_WGS_PER_CU[1]) if block_rows == 6 else (cus * 2, cus * 2)
    if _hook("VGGSFM_TILE_WGS"):                       # measurement hook: workgroups per CU as floats, "off,diag"
        fo, fd = (float(x) for x in _hook("VGGSFM_TILE_WGS").split(","))
        slots = (max(1, int(cus * fo)), max(1, int(cus * fd)))
    # three batches when the factorisation can overlap the later ones (enough camera groups, enough work per batch)
    overlap = OVERLAP_FACTORIZATION if overlap is None else bool(overlap)
    nb = TILE_BATCHES if (overlap and int(obs_cam.shape[0]) >= OVERLAP_MIN_OBS and S >= OVERLAP_MIN_FRAMES) else 1
    # small problems: one tile launch (c2, 0.25 M observations: 0.101 -> 0.068 ms for the tiles; c3, 5 M: 0.84 -> 0.90 ms)
    merged = nb == 1 and int(obs_cam.shape[0]) < MERGED_TILE_MAX_OBS
    merged_slots = slots[0] if merged else None
    backfill = TILE_BACKFILL if block_rows == 6 else None
    if _hook("VGGSFM_TILE_BACKFILL"):                  # measurement hook: "off,diag" workgroups per CU, or "0" = two launches
        v = _hook("VGGSFM_TILE_BACKFILL")
        backfill = None if v == "0" else tuple(float(x) for x in v.split(","))
    if backfill is not None and nb == 1 and not merged:
        # ONE launch, off-diagonal chunks first (one resident round at the merged kernel's occupancy), the diagonal chunks
        # behind them: the in-order dispatcher hands a diagonal workgroup to every slot an off-diagonal one leaves, so the
        # ~16 % of the chip that idled while the slowest off-diagonal tiles finished (round 3 phase trace) computes diagonal
        # tiles instead.  (Not the round-2 "merged" form, where both kinds shared one resident round and swept the points
        # together: 0.90 ms against 0.62 + 0.24.)
        merged, merged_slots = True, None
        slots = (max(1, int(cus * backfill[0])), max(1, int(cus * backfill[1])))
    chunk_desc, entries, tile_desc, obs_slot, nseg, batch_desc = build_schur_tiles(
        row_ptr, obs_cam, max_chunks=slots, num_batches=nb, later_scale=(cus - CHOL_CUS) / cus,
        merged_slots=merged_slots, block_rows=block_rows, simple=int(obs_cam.shape[0]) < SIMPLE_WORKLIST_MAX_OBS, num_cams=S)
    prob = DeviceProblem(cam_q, cam_t, intr, pts, row_ptr, obs_cam, obs_uv, col_ptr, cobs_pt, cobs_uv, chunk_desc,
                         entries, tile_desc, obs_slot, nseg, MODEL_ID[camera_type], cam_const=cam_const,
                         batch_desc=batch_desc, chol_split=chol_split, cam_perm=cam_perm, merged_tile_launch=merged,
                         refine_focal=bool(refine_focal_length), refine_extra=bool(refine_extra_params))
    if first_group is not None:
        kd_max = 2 if camera_type == "SIMPLE_RADIAL" else 1          # upper bound of the intrinsics unknowns per block
        prob.chol_first_blk = envelope_blocks(first_group, S, 6 * S + kd_max * n_intr).to(dev)
    return prob, valid_idx, deleted


def window_bundle_adjustment(points3d, extrinsics, intrinsics, tracks, masks, num_existing_points, extra_params=None,
                             shared_camera=True, camera_type="SIMPLE_RADIAL", options=None):
    """The local BA of one video window (vggsfm/runners/video_runner.py:800-838): frame 0 of the window (= the last
    frame of the previous one) keeps its pose, the first `num_existing_points` VALID points (the ones carried
    over from earlier windows) are constant, the newly triangulated ones are variable, focal length and
    distortion are not refined (``ba_options.refine_focal_length = refine_extra_params = False``)."""
    options = copy.deepcopy(options) if options is not None else BundleAdjustmentOptions()   # (the caller's object stays as it is)
    options.refine_focal_length = False
    options.refine_extra_params = False
    valid = masks.bool().sum(0) >= 2
    order = torch.cumsum(valid.long(), 0) - 1                       # 0-based point3D id - 1 of every valid track
    constant = valid & (order < num_existing_points)
    return bundle_adjustment(points3d, extrinsics, intrinsics, tracks, masks, None, extra_params, shared_camera,
                             camera_type, options, False, constant_points=constant, constant_pose_frames=[0],
                             filter_negative_depth=False)


def _c_options(options: BundleAdjustmentOptions, overlap=True):
    so = options.solver_options
    return _lib.BAOptions(so.max_num_iterations, so.max_num_consecutive_invalid_steps, int(so.jacobi_scaling),
                          so.function_tolerance, so.gradient_tolerance, so.parameter_tolerance,
                          so.initial_trust_region_radius, so.max_trust_region_radius, so.min_trust_region_radius,
                          so.min_lm_diagonal, so.max_lm_diagonal, so.min_relative_decrease,
                          CHOL_CUS if (overlap and OVERLAP_FACTORIZATION) else 0)


def _summary_dict(summ, log, n):
    its = [dict(iteration=l.iteration, cost=l.cost, cost_change=l.cost_change, gradient_max_norm=l.gradient_max_norm,
                step_norm=l.step_norm, relative_decrease=l.relative_decrease, radius=l.radius,
                successful=bool(l.successful)) for l in log[:n]]
    return dict(initial_cost=summ.initial_cost, final_cost=summ.final_cost, num_iterations=summ.num_iterations,
                num_successful_steps=summ.num_successful_steps, num_unsuccessful_steps=summ.num_unsuccessful_steps,
                termination=summ.termination, termination_str=TERMINATION.get(summ.termination, "?"),
                n_reduced=summ.n_reduced, iterations=its)


def solve(problem: DeviceProblem, options: Optional[BundleAdjustmentOptions] = None, workspace=None):
    """Run the LM loop of `problem` in place on the current stream; one host sync at the end."""
    L = _lib.lib()
    options = options or BundleAdjustmentOptions()
    problem.refine_focal = options.refine_focal_length
    problem.refine_extra = options.refine_extra_params
    problem.loss = LOSS_ID[options.loss_function_type]
    problem.loss_scale = options.loss_function_scale
    cp = problem.c_struct()
    co = _c_options(options)
    nbytes = int(L.vgg_ba_workspace_bytes(ctypes.byref(cp), ctypes.byref(co)))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=problem.pts.device)
    summ = _lib.BASummary()
    cap = options.solver_options.max_num_iterations + 2
    log = (_lib.BAIteration * cap)()
    rc = L.vgg_ba_solve(ctypes.byref(cp), ctypes.byref(co), _lib.ptr(workspace), ctypes.c_size_t(workspace.numel()),
                        ctypes.byref(summ), log, cap, _lib.stream_ptr())
    _lib.check(rc, "vgg_ba_solve")
    out = _summary_dict(summ, log, summ.num_log)
    if options.print_summary:
        print(f"Bundle adjustment report: residuals {problem.num_obs * 2}, parameters reduced {summ.n_reduced}, "
              f"iterations {summ.num_iterations}, cost {summ.initial_cost:.6g} -> {summ.final_cost:.6g}, "
              f"{out['termination_str']}")
    return out, workspace


def normalization_transform(extrinsics, extent=5.0, p0=0.1, p1=0.9):
    """The similarity of Reconstruction.normalize(extent, p0, p1, True): (scale (), mean (3,), normalised extrinsics) --
    X -> scale (X - mean); None when there are fewer than two cameras."""
    S = extrinsics.shape[0]
    if S < 2:
        return None
    R, t = extrinsics[:, :, :3], extrinsics[:, :, 3]
    centers = -torch.einsum("sji,sj->si", R, t)
    c32, _ = torch.sort(centers.to(torch.float32), dim=0)
    P0 = int(p0 * (S - 1)) if S > 3 else 0
    P1 = int(p1 * (S - 1)) if S > 3 else S - 1
    bmin, bmax = c32[P0].double(), c32[P1].double()
    mean = c32[P0:P1 + 1].double().sum(0) / (P1 - P0 + 1)
    old_extent = (bmax - bmin).norm()
    scale = torch.where(old_extent < torch.finfo(torch.float64).eps, torch.ones_like(old_extent), extent / old_extent)
    ext = extrinsics.clone()
    ext[:, :, 3] = scale * (t + torch.einsum("sij,j->si", R, mean))
    return scale, mean, ext


def normalize_reconstruction(extrinsics, points3D, alive=None, extent=5.0, p0=0.1, p1=0.9):
    """Reconstruction.normalize(5.0, 0.1, 0.9, True) (vggsfm/utils/triangulation.py:1217) on tensors."""
    tr = normalization_transform(extrinsics, extent, p0, p1)
    if tr is None:
        return extrinsics, points3D
    scale, mean, ext = tr
    pts = scale * (points3D - mean)
    if alive is not None:          # (a select, not a masked assignment: no index list, no host synchronisation on the device)
        pts = torch.where(alive[:, None], pts, points3D)
    return ext, pts


def bundle_adjustment(points3d, extrinsics, intrinsics, tracks, masks, image_size=None, extra_params=None,
                      shared_camera=False, camera_type="SIMPLE_PINHOLE", options=None, normalize=False,
                      constant_points=None, constant_pose_frames=None, filter_negative_depth=True):
    """Tensor-in / tensor-out equivalent of the reference's three-call round trip
    (batch_matrix_to_pycolmap -> pycolmap.bundle_adjustment -> pycolmap_to_batch_matrix).
    Returns (points3D_opt (P',3), extrinsics (S,3,4), intrinsics (S,3,3), extra_params (S,1)|None, summary);
    P' = number of tracks with >= 2 masked observations, rows of deleted points are zero.

    The two optional arguments express a ``pycolmap.BundleAdjustmentConfig`` (video_runner.py:813-829):
    `constant_points` (P,) bool over the INPUT tracks = ``add_constant_point``; `constant_pose_frames` = the frames
    of ``set_constant_cam_pose`` -- when given it REPLACES the default gauge of ``pycolmap.bundle_adjustment``
    (frame 0 pose + frame 1 t_x constant).  `filter_negative_depth=False` for the BundleAdjuster-level entry
    (``solve_bundle_adjustment``), which does not run the ObservationManager filter."""
    _lib.require_gpu(points3d, extrinsics, intrinsics, tracks, masks)
    options = options or BundleAdjustmentOptions()
    sort_points = _hook("VGGSFM_SORT_POINTS", "1" if SORT_POINTS else "0") != "0"
    prob, valid_idx, deleted = compile_problem(points3d, extrinsics, intrinsics, tracks, masks, extra_params,
                                               shared_camera, camera_type, filter_negative_depth=filter_negative_depth,
                                               gauge="colmap" if constant_pose_frames is None else "config",
                                               camera_split=True, refine_focal_length=options.refine_focal_length,
                                               refine_extra_params=options.refine_extra_params, sort_points=sort_points)
    S = extrinsics.shape[0]
    inv_perm = None
    if prob.cam_perm is not None:
        inv_perm = torch.empty_like(prob.cam_perm)
        inv_perm[prob.cam_perm] = torch.arange(S, device=prob.cam_perm.device)
    if constant_pose_frames is not None:
        cf = torch.as_tensor(list(constant_pose_frames), dtype=torch.long, device=prob.cam_const.device)
        prob.cam_const[cf if inv_perm is None else inv_perm[cf]] = 1
    if constant_points is not None:
        prob.pt_const = constant_points.to(device=prob.pts.device)[valid_idx].to(torch.uint8).contiguous()
    summary, _ = solve(prob, options)
    ext = torch.cat([quat_to_rotmat(prob.cam_q), prob.cam_t[:, :, None]], -1)
    if inv_perm is not None:
        ext = ext[inv_perm]                           # back to the order of the input frames
    pts = prob.pts
    pts[deleted] = 0.0
    if sort_points:                                   # back to track order (the contract: rows of the valid tracks, ascending)
        back = torch.argsort(valid_idx)
        pts, deleted, valid_idx = pts[back], deleted[back], valid_idx[back]
    if normalize:
        ext, pts = normalize_reconstruction(ext, pts, ~deleted)
    idx = torch.zeros(S, dtype=torch.long, device=pts.device) if shared_camera else \
        (torch.arange(S, device=pts.device) if inv_perm is None else inv_perm)
    K = torch.zeros((S, 3, 3), dtype=torch.float64, device=pts.device)
    K[:, 0, 0] = K[:, 1, 1] = prob.intr[idx, 0]
    K[:, 0, 2] = prob.intr[idx, 1]
    K[:, 1, 2] = prob.intr[idx, 2]
    K[:, 2, 2] = 1.0
    extra = prob.intr[idx, 3][:, None].clone() if camera_type == "SIMPLE_RADIAL" else None
    summary["valid_idx"] = valid_idx
    summary["deleted"] = deleted
    return pts, ext, K, extra, summary
