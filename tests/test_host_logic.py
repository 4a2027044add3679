"""CPU tests of the host logic (torch ops that compile the BA problem / Schur work list) and of the
C-ABI library's exports.  No compute kernel is called here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import ba as OB
from vggsfm_amd import _lib
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene, perturb_for_ba

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _full_work_list(monkeypatch):
    """The scenes of this file are small; the scheduling invariants they check are those of the FULL work-list construction
    (density order, pattern grouping, cost model).  `test_simple_work_list_*` switches the small-problem path back on."""
    monkeypatch.setattr(BA, "SIMPLE_WORKLIST_MAX_OBS", 0)


def T(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x))


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "vggsfm_amd.h")).read()
    declared = set(re.findall(r"\b(vgg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTED)
    for sym in declared:
        assert hasattr(L, sym), sym
    L.vgg_build_arch.restype = ctypes.c_char_p
    assert L.vgg_build_arch() == b"gfx950"
    # the binding refuses a library with another struct layout: version and sizes are compared at load
    assert int(re.search(r"#define VGG_ABI_VERSION (\d+)", header).group(1)) == _lib.ABI_VERSION == L.vgg_abi_version()
    L.vgg_abi_sizeof.restype = ctypes.c_size_t
    assert [int(L.vgg_abi_sizeof(i)) for i in range(5)] == [ctypes.sizeof(t) for t in (_lib.BAProblem, _lib.BAOptions,
                                                                                       _lib.BAIteration, _lib.BASummary)] + [0]
    assert _lib.lib() is not None


def test_binding_rejects_a_library_with_another_abi(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.lib()


def test_product_path_has_no_cpu_fallback():
    from vggsfm_amd.utils import triangulation_helpers as H
    with pytest.raises(RuntimeError, match="no CPU path"):
        H.project_3D_points(torch.zeros(4, 3, dtype=torch.float64), torch.zeros(2, 3, 4, dtype=torch.float64),
                            torch.zeros(2, 3, 3, dtype=torch.float64))
    # and nothing under vggsfm_amd/ imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vggsfm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} references the oracle"


def test_quaternion_conversion_matches_oracle():
    rng = np.random.default_rng(0)
    from scipy.spatial.transform import Rotation
    R = Rotation.random(200, random_state=3).as_matrix()
    R[0] = np.diag([1.0, -1.0, -1.0])      # trace < 0 branches
    R[1] = np.diag([-1.0, 1.0, -1.0])
    R[2] = np.diag([-1.0, -1.0, 1.0])
    q = BA.rotmat_to_quat(T(R)).numpy()
    np.testing.assert_allclose(q, OB.rotmat_to_quat(R), atol=1e-15)
    np.testing.assert_allclose(BA.quat_to_rotmat(T(q)).numpy(), R, atol=1e-14)


def _check_views_and_work_list(prob, S, vi, row_ptr, obs_cam):
    """Invariants of a compiled problem: the camera-major view holds the point-major observations, and the Schur work
    list (chunks -> entries -> segments) covers every co-observing camera pair of every point exactly once."""
    # camera-major view holds the same observations
    cm = set()
    cp = prob.col_ptr.numpy()
    for c in range(S):
        for j in range(cp[c], cp[c + 1]):
            cm.add((c, int(prob.cobs_pt[j]), float(prob.cobs_uv[j, 0]), float(prob.cobs_uv[j, 1])))
    pm = set()
    for p in range(len(vi)):
        for o in range(row_ptr[p], row_ptr[p + 1]):
            pm.add((int(obs_cam[o]), p, float(prob.obs_uv[o, 0]), float(prob.obs_uv[o, 1])))
    assert cm == pm
    # Schur work list covers every co-observing camera pair of every point exactly once
    desc, ent = prob.chunk_desc.numpy(), prob.entries.numpy()
    slot = prob.obs_slot.numpy()
    assert len(np.unique(slot)) == len(slot) and slot.max() < prob.num_segments * BA.GROUP
    assert (slot % BA.GROUP == obs_cam % BA.GROUP).all()
    seg_of = slot // BA.GROUP
    seg_first = {int(sg): int(np.nonzero(seg_of == sg)[0][0]) for sg in np.unique(seg_of)}
    seg_mask = {int(sg): sum(1 << (int(c) % BA.GROUP) for c in obs_cam[seg_of == sg]) for sg in np.unique(seg_of)}
    pt_of_obs = np.repeat(np.arange(len(vi)), np.diff(row_ptr))
    has = np.diff(row_ptr) > 0
    ncam = int(obs_cam.max()) + 1
    key = np.where(has, obs_cam[np.minimum(row_ptr[:-1], len(obs_cam) - 1)].astype(np.int64) * ncam
                   + obs_cam[np.maximum(row_ptr[1:] - 1, 0)], 0)
    sweep_rank = np.empty(len(vi), np.int64)
    sweep_rank[np.argsort(key, kind="stable")] = np.arange(len(vi))
    covered = {}
    seen_entries = np.zeros(len(ent), bool)
    pos_of_point = {}
    for gI, gJ, tb, te, j, J in desc:
        # workgroup j of J of the tile sweeps the strided sub-chunks j, j+J, ... of SUB entries
        assert gI <= gJ and 0 <= j < J and J <= max(1, -(-(te - tb) // BA.MIN_CHUNK))
        # inside a tile: one entry per point; ascending sweep positions from window to window (QUAD_SORT_WINDOW entries), and
        # inside a window grouped by the pattern of 16-row blocks the tile kernel can skip, ascending inside a group
        # (field 0 of an entry = its point -- the diagonal tile launch looks the point's back-substitution block up by it;
        #  the sweep position of a point = its rank by (first camera, last camera), ties in point order)
        pt_t = ent[tb:te, 0].astype(np.int64)
        assert len(np.unique(pt_t)) == len(pt_t)
        pos_t = sweep_rank[pt_t]
        bd = 6 if prob.intr.shape[0] == 1 else 6 + (2 if prob.camera_model == BA.MODEL_ID["SIMPLE_RADIAL"] else 1)
        W = BA.QUAD_SORT_WINDOW if BA.QUAD_SORT_WINDOW > 0 else 1
        for w0 in range(0, te - tb, W):
            blk = pos_t[w0:w0 + W]
            if w0:
                assert blk.min() > pos_t[w0 - W:w0].max()
            pats = [(_block_pattern(seg_mask[sa], bd), _block_pattern(seg_mask[sb], bd)) for sa, sb in ent[tb + w0:tb + w0 + len(blk), 1:3]]
            assert pats == sorted(pats)
            for a in range(1, len(blk)):
                assert pats[a] != pats[a - 1] or blk[a] > blk[a - 1]
        own = [k for s0 in range(tb + j * BA.SUB, te, J * BA.SUB) for k in range(s0, min(s0 + BA.SUB, te))]
        for k in own:
            assert not seen_entries[k]
            seen_entries[k] = True
            pos, sa, sb, msk = ent[k]
            mA, mB = seg_mask[sa], seg_mask[sb]
            # field 3 = the union of the own masks over the quad the entry belongs to (aligned to the tile's begin)
            q0 = tb + (k - tb) // 4 * 4
            quad = 0
            for kk in range(q0, min(q0 + 4, te)):
                quad |= seg_mask[ent[kk, 1]] | (seg_mask[ent[kk, 2]] << 16)
            assert (int(msk) & 0xffffffff) == quad
            ca, cb = bin(mA).count("1"), bin(mB).count("1")
            oa, ob = seg_first[sa], seg_first[sb]
            A = obs_cam[oa:oa + ca]
            B = obs_cam[ob:ob + cb]
            assert (A // BA.GROUP == gI).all() and (B // BA.GROUP == gJ).all()
            p = int(pt_of_obs[oa])
            assert int(pos) == p                                 # field 0 is the point of both segments
            pos_of_point[p] = int(sweep_rank[p])
            assert row_ptr[p] <= oa and oa + ca <= row_ptr[p + 1] and row_ptr[p] <= ob and ob + cb <= row_ptr[p + 1]
            for a in A:
                for bb in B:
                    if gI == gJ and a > bb:
                        continue
                    covered[(p, int(a), int(bb))] = covered.get((p, int(a), int(bb)), 0) + 1
    # the sweep order: points by (first camera, last camera)
    by_pos = sorted(pos_of_point, key=pos_of_point.get)
    keys = [(int(obs_cam[row_ptr[q]]), int(obs_cam[row_ptr[q + 1] - 1])) for q in by_pos]
    assert keys == sorted(keys) and len(set(pos_of_point.values())) == len(pos_of_point)
    assert seen_entries.all()
    assert len(desc) <= max(2 * 256 * 4, len(prob.tile_desc))  # one resident round per launch (off-diag, diag)
    # tiles: consecutive chunk ranges that cover all chunks exactly once, one tile per (gI, gJ)
    td = prob.tile_desc.numpy()
    assert td[0, 2] == 0 and td[-1, 3] == len(desc) and (td[1:, 2] == td[:-1, 3]).all()
    assert len({(int(a), int(b)) for a, b, _, _ in td}) == len(td)
    for gI, gJ, c0, c1 in td:
        assert (desc[c0:c1, 0] == gI).all() and (desc[c0:c1, 1] == gJ).all()
    expect = set()
    for p in range(len(vi)):
        cams = obs_cam[row_ptr[p]:row_ptr[p + 1]]
        for i, a in enumerate(cams):
            for bb in cams[i:]:
                expect.add((p, int(a), int(bb)))
    assert set(covered) == expect and all(v == 1 for v in covered.values())

def _block_pattern(mask, bd=6, group=16):
    """bit b <=> a camera of the segment has rows in the 16-row block b of the tile side (what schur_tile_kernel skips by)"""
    out = 0
    for b in range(bd * group // 16):
        s0, s1 = (16 * b) // bd, min(group - 1, (16 * b + 15) // bd)
        if int(mask) & (((2 << s1) - 1) & ~((1 << s0) - 1)):
            out |= 1 << b
    return out



@pytest.mark.parametrize("density_cut", [0.0, 2.0])    # in-place filtering of the grid / the observation-list construction
@pytest.mark.parametrize("S,N", [(5, 40), (40, 300), (70, 150)])
def test_compile_problem_matches_oracle_construction(S, N, density_cut, monkeypatch):
    monkeypatch.setattr(BA, "SPARSE_GRID_DENSITY", density_cut)
    monkeypatch.setattr(BA, "SMALL_GRID_CELLS", 0)            # (small grids skip the fill test: these scenes are small)
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=S)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=S)
    masks = sc.mask.copy()
    masks[:, 3] = False
    masks[0, 3] = True                     # single observation: not a point
    pts0[5] = [0, 0, -2.0]                 # behind everything
    pts0[6, 0] = 4000.0                    # >= 3000
    masks[2:, 7] = False
    pts0[7] = -pts0[7]
    prob, valid_idx, deleted = BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(masks), T(extra0), True,
                                                  "SIMPLE_RADIAL")
    vi, row_ptr, obs_cam, obs_uv, dele = OB.build_observations(pts0, ext0, sc.tracks, masks)
    assert np.array_equal(valid_idx.numpy(), vi)
    assert np.array_equal(deleted.numpy(), dele)
    assert np.array_equal(prob.row_ptr.numpy(), row_ptr)
    assert np.array_equal(prob.obs_cam.numpy(), obs_cam)
    assert np.array_equal(prob.obs_uv.numpy().astype(np.float64), obs_uv)
    _check_views_and_work_list(prob, S, vi, row_ptr, obs_cam)


@pytest.mark.parametrize("density_cut", [0.0, 2.0])
@pytest.mark.parametrize("S,N", [(5, 40), (40, 300), (70, 150)])
def test_compile_problem_sorted_points_is_the_same_problem_renumbered(S, N, density_cut, monkeypatch):
    """`sort_points=True` (what ba.bundle_adjustment compiles with) numbers the points by track length.  Both
    constructions (grid / observation list) must give the SAME problem up to that renumbering: same valid tracks, same deleted
    flags per track, and per point the same (camera, pixel) list in camera order; per camera the same SET of (point, pixel)
    -- the order inside a camera is the list's, not ascending by point, and nothing needs it to be."""
    monkeypatch.setattr(BA, "SPARSE_GRID_DENSITY", density_cut)
    monkeypatch.setattr(BA, "SMALL_GRID_CELLS", 0)            # (small grids skip the fill test: these scenes are small)
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=S)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=S)
    masks = sc.mask.copy()
    masks[:, 3] = False
    masks[0, 3] = True
    pts0[5] = [0, 0, -2.0]
    pts0[6, 0] = 4000.0
    args = (T(pts0), T(ext0), T(K0), T(sc.tracks), T(masks), T(extra0), True, "SIMPLE_RADIAL")
    pa, va, da = BA.compile_problem(*args)
    pb, vb, db = BA.compile_problem(*args, sort_points=True)
    va, vb = va.numpy(), vb.numpy()
    assert np.array_equal(np.sort(vb), va)
    lengths = np.diff(pb.row_ptr.numpy())
    raw = masks[:, vb].sum(0)
    assert (np.diff(raw) >= 0).all()                       # numbered by (input) track length, ascending
    back = np.argsort(vb)                                  # sorted id of track-order point i
    assert np.array_equal(db.numpy()[back], da.numpy())
    assert np.array_equal(pb.pts.numpy()[back], pa.pts.numpy())
    ra, rb = pa.row_ptr.numpy(), pb.row_ptr.numpy()
    for i in range(len(va)):
        j = back[i]
        assert np.array_equal(pa.obs_cam.numpy()[ra[i]:ra[i + 1]], pb.obs_cam.numpy()[rb[j]:rb[j + 1]])
        assert np.array_equal(pa.obs_uv.numpy()[ra[i]:ra[i + 1]], pb.obs_uv.numpy()[rb[j]:rb[j + 1]])
    assert lengths.sum() == np.diff(ra).sum()
    ca, cb = pa.col_ptr.numpy(), pb.col_ptr.numpy()
    assert np.array_equal(ca, cb)
    for c in range(S):
        A = sorted((int(p), tuple(uv)) for p, uv in zip(pa.cobs_pt.numpy()[ca[c]:ca[c + 1]], pa.cobs_uv.numpy()[ca[c]:ca[c + 1]]))
        Bq = sorted((int(np.nonzero(back == p)[0][0]), tuple(uv))
                    for p, uv in zip(pb.cobs_pt.numpy()[cb[c]:cb[c + 1]], pb.cobs_uv.numpy()[cb[c]:cb[c + 1]]))
        assert A == Bq
    vi, row_ptr, obs_cam, _, _ = OB.build_observations(pts0, ext0, sc.tracks, masks)
    assert np.array_equal(va, vi)


@pytest.mark.parametrize("order", ["plain", "sparse_first", "dense_first", "stride", "sdm"])
@pytest.mark.parametrize("top_up", [False, True])
def test_tile_scheduling_choices_keep_the_work_list_invariants(order, top_up, monkeypatch):
    """Round-5 scheduling of the Schur tile work list (ba.build_schur_tiles): the launch order of the tiles (by density), the
    top-up of the slots the chunk-size search leaves empty and the per-launch fixed cost only decide WHERE and by how many
    workgroups a tile is processed -- every order covers every co-observing camera pair of every point exactly once, with
    the same entries per tile, and a launch never gets more workgroups than resident slots."""
    monkeypatch.setattr(BA, "TILE_ORDER", order)
    monkeypatch.setattr(BA, "TILE_TOP_UP", top_up)
    if order in BA.TILE_ORDERS_EXPERIMENTAL:               # (the experimental orders are reachable through the hook gate only)
        monkeypatch.setenv("VGGSFM_AMD_DEBUG_HOOKS", "1")
        monkeypatch.setenv("VGGSFM_TILE_ORDER", order)
    S, N = 70, 400
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=7)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=7)
    prob, valid_idx, _ = BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL")
    vi, row_ptr, obs_cam, _, _ = OB.build_observations(pts0, ext0, sc.tracks, sc.mask)
    _check_views_and_work_list(prob, S, vi, row_ptr, obs_cam)
    desc, ent, tiles = prob.chunk_desc.numpy(), prob.entries.numpy(), prob.tile_desc.numpy()
    # the tiles partition the chunks in launch order, off-diagonal tiles in front of the diagonal ones
    assert tiles[0, 2] == 0 and (tiles[1:, 2] == tiles[:-1, 3]).all() and tiles[-1, 3] == len(desc)
    is_diag = tiles[:, 0] == tiles[:, 1]
    assert not (is_diag[:-1] & ~is_diag[1:]).any()
    # same entries per tile as the (gI, gJ) order
    monkeypatch.setattr(BA, "TILE_ORDER", "plain")
    monkeypatch.setattr(BA, "TILE_TOP_UP", False)
    ref, _, _ = BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL")
    per_tile = lambda pr: {(int(a), int(b)): sorted(map(tuple, pr.entries.numpy()[tb:te, :3].tolist()))
                           for a, b, tb, te in {(r[0], r[1], r[2], r[3]) for r in pr.chunk_desc.numpy().tolist()}}
    assert per_tile(prob) == per_tile(ref)
    cus = 256
    off = int((desc[:, 0] != desc[:, 1]).sum())
    assert len(desc) <= cus * BA.TILE_WGS_PER_CU[0] or (off <= cus * BA.TILE_WGS_PER_CU[0] and len(desc) - off <= cus * BA.TILE_WGS_PER_CU[1])


@pytest.mark.parametrize("density_cut", [0.0, 2.0])
@pytest.mark.parametrize("S,N,density,seed", [(2, 30, 1.0, 0), (3, 50, 0.7, 1), (16, 80, 0.3, 2), (17, 120, 0.15, 3), (33, 200, 0.08, 4),
                                              (48, 60, 0.9, 5), (40, 150, 0.05, 6)])
def test_compile_problem_on_irregular_visibility(S, N, density, seed, density_cut, monkeypatch):
    """The same invariants on visibility patterns make_scene does not produce: Bernoulli masks (tracks with gaps), cameras
    that see nothing, points seen by exactly two cameras of different groups, frame counts around the 16-camera group
    size, per-frame intrinsics."""
    monkeypatch.setattr(BA, "SPARSE_GRID_DENSITY", density_cut)
    monkeypatch.setattr(BA, "SMALL_GRID_CELLS", 0)            # (small grids skip the fill test: these scenes are small)
    rng = np.random.default_rng(seed)
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=False, seed=seed)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=seed)
    masks = rng.random((S, N)) < density
    if S > 4:
        masks[rng.integers(2, S)] = False                   # a camera that sees nothing
        masks[:, 0] = False
        masks[0, 0] = masks[S - 1, 0] = True                # a point seen by the first and the last camera only
    # keep the points in front of the cameras that observe them irrelevant: the construction only looks at masks, depth
    # and coordinates; use the scene's own (valid) geometry and let compile_problem / the oracle filter alike
    prob, valid_idx, deleted = BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(masks), T(extra0), False,
                                                  "SIMPLE_RADIAL")
    vi, row_ptr, obs_cam, obs_uv, dele = OB.build_observations(pts0, ext0, sc.tracks, masks)
    assert np.array_equal(valid_idx.numpy(), vi) and np.array_equal(deleted.numpy(), dele)
    assert np.array_equal(prob.row_ptr.numpy(), row_ptr) and np.array_equal(prob.obs_cam.numpy(), obs_cam)
    if len(vi) == 0:
        return
    _check_views_and_work_list(prob, S, vi, row_ptr, obs_cam)


def test_normalize_matches_oracle():
    sc = make_scene(12, 50, "SIMPLE_PINHOLE", seed=2)
    alive = np.ones(50, bool)
    alive[3] = False
    e1, p1 = BA.normalize_reconstruction(T(sc.extrinsics), T(sc.points3D), T(alive))
    e2, p2 = OB.normalize_reconstruction(sc.extrinsics, sc.points3D, alive)
    np.testing.assert_allclose(e1.numpy(), e2, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(p1.numpy(), p2, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("num_batches", [1, 2, 3])
@pytest.mark.parametrize("max_chunks", [1, 7, 40, 100000])
def test_schur_chunks_adaptive_cover(max_chunks, num_batches):
    """Adaptive chunk size: every entry belongs to exactly one workgroup's strided sub-chunks and the
    workgroup count respects the device capacity (or is one per tile when there are more tiles than slots).
    Batches: consecutive chunk / tile ranges, off-diagonal chunks before the diagonal ones inside each batch,
    camera groups gI non-decreasing over the batches, and every tile of a batch has gI >= first_camera_group."""
    torch.manual_seed(0)
    S, P = 40, 300
    m = torch.rand(S, P) < 0.4
    m[:2] = True
    pm = torch.nonzero(m.t())
    obs_cam = pm[:, 1].to(torch.int32)
    row_ptr = torch.zeros(P + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(m.sum(0), 0).to(torch.int32)
    desc, ent, tiles, slot, nseg, batches = BA.build_schur_tiles(row_ptr, obs_cam, max_chunks=max_chunks,
                                                                 num_batches=num_batches, later_scale=0.875)
    seen = np.zeros(len(ent), int)
    for gI, gJ, tb, te, j, J in desc.numpy():
        for s0 in range(tb + j * BA.SUB, te, J * BA.SUB):
            seen[s0:min(s0 + BA.SUB, te)] += 1
    assert (seen == 1).all()
    assert len(desc) <= max(2 * max_chunks * num_batches, len(tiles))
    td, cd, bd = tiles.numpy(), desc.numpy(), batches.numpy()
    assert td[0, 2] == 0 and td[-1, 3] == len(desc) and (td[1:, 2] == td[:-1, 3]).all()
    assert not batches.is_cuda and bd.shape == (num_batches, 6)
    assert bd[0, 0] == 0 and bd[-1, 2] == len(cd) and bd[0, 3] == 0 and bd[-1, 4] == len(td)
    assert (bd[1:, 0] == bd[:-1, 2]).all() and (bd[1:, 3] == bd[:-1, 4]).all() and (np.diff(bd[:, 5]) >= 0).all()
    for c0, cm, c1, t0, t1, g0 in bd:
        assert (cd[c0:cm, 0] != cd[c0:cm, 1]).all() and (cd[cm:c1, 0] == cd[cm:c1, 1]).all()
        assert (td[t0:t1, 0] >= g0).all() and (td[t0:t1, 2] >= c0).all() and (td[t0:t1, 3] <= c1).all()
        if t1 > t0:
            assert td[t0:t1, 0].min() == g0
    # a camera group belongs to one batch only (column blocks of the reduced system complete batch by batch)
    owner = {}
    for b, (c0, cm, c1, t0, t1, g0) in enumerate(bd):
        for g in np.unique(td[t0:t1, 0]):
            assert owner.setdefault(int(g), b) == b


def test_generate_combinations_matches_itertools_order():
    """vggsfm/utils/triangulation_helpers.py:638-645 uses itertools.combinations; the vectorised mirror must give
    the same pairs in the same order (the order selects the RANSAC hypotheses)."""
    import itertools

    from vggsfm_amd.utils.triangulation_helpers import generate_combinations
    for n in (2, 3, 8, 30, 200):
        ref = np.array(list(itertools.combinations(np.arange(n), 2))).reshape(-1, 2)
        got = generate_combinations(n)
        assert got.dtype == ref.dtype and np.array_equal(got, ref)


def test_find_camera_split_on_sliding_window_visibility():
    """Video-like visibility (tracks span at most 40 % of the frames): the first and the last cameras share no point, the
    ordering [A, B, rest] makes the leading part of the reduced system block diagonal, A's column count is a multiple of
    64 and frames 0 / 1 (the gauge) stay in front.  All-to-all visibility: no split."""
    sc = make_scene(200, 4000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
    m = T(sc.mask)
    perm, (ca, cb) = BA.find_camera_split(m)
    assert perm is not None and ca % 64 == 0 and ca > 0 and cb > 0 and min(ca, cb) // 64 >= BA.CAMERA_SPLIT_MIN_STEPS
    assert sorted(perm.tolist()) == list(range(200)) and perm[:2].tolist() == [0, 1]
    na, nb_ = ca // 6, cb // 6
    mp = m[perm]
    assert not bool((mp[:na].any(0) & mp[na:na + nb_].any(0)).any())          # no point seen from both A and B
    assert (ca, cb) == (384, 288) and perm[-8:].tolist() == list(range(192, 200))     # groups 0-3 vs 9-11; the partial group last
    # a reduce callback that adds a coupling (another rank sees a point from the first and the last group) kills it
    def couple(adj):
        adj[0, -1] = adj[-1, 0] = 1.0
    assert BA.find_camera_split(m, adjacency_reduce=couple)[0] is None
    full = torch.ones(200, 50, dtype=torch.bool)
    assert BA.find_camera_split(full)[0] is None and BA.find_camera_split(m[:40])[0] is None


@pytest.mark.parametrize("seed", range(8))
def test_camera_split_and_order_decouple_on_random_banded_visibility(seed):
    """Random sequences: S frames, tracks of random length up to a random band, random frame counts (not multiples of the
    16-camera group).  Whatever the two routines return is either None or a valid structure: a permutation with the
    gauge frames first; for the split, no point seen from both leading blocks and A a multiple of 64 columns; for the
    k-way order, an envelope that no co-visible camera pair violates."""
    rng = np.random.default_rng(seed)
    S = int(rng.integers(70, 420))
    band = int(rng.integers(8, max(9, S // 3)))
    P = 3000
    start = rng.integers(0, S - 3, P)
    ln = rng.integers(3, band + 1, P)
    fr = np.arange(S)[:, None]
    m = torch.from_numpy((fr >= start[None]) & (fr < (start + ln)[None]))
    perm, (ca, cb) = BA.find_camera_split(m)
    if perm is not None:
        assert sorted(perm.tolist()) == list(range(S)) and perm[:2].tolist() == [0, 1]
        assert ca % 64 == 0 and ca > 0 and cb > 0 and ca % 6 == 0 and cb % 6 == 0
        mp = m[perm]
        na, nb_ = ca // 6, cb // 6
        assert not bool((mp[:na].any(0) & mp[na:na + nb_].any(0)).any())
    perm2, fg = BA.find_camera_order(m)
    if perm2 is not None:
        assert sorted(perm2.tolist()) == list(range(S)) and perm2[:2].tolist() == [0, 1]
        fb = BA.envelope_blocks(fg, S, 6 * S + 2)
        mp = m[perm2].float()
        covis = (mp @ mp.t()) > 0
        first_cam = torch.tensor([int(torch.nonzero(covis[c])[0]) if bool(covis[c].any()) else c for c in range(S)])
        first_col = 6 * first_cam[torch.arange(6 * S) // 6]
        for r in range(0, 6 * S, 64):
            assert int(fb[r // 64]) * 64 <= int(first_col[r:r + 64].min())


def test_find_camera_order_kway_and_envelope():
    """Video-like visibility (1000 frames, tracks of at most 40 frames): ``find_camera_order`` returns a nested-dissection
    order -- k > 2 interior runs, then their separators -- in which the interiors do not couple, frames 0 / 1 (the gauge)
    stay in front, and ``envelope_blocks`` is a valid ROW ENVELOPE of the permuted reduced system: no co-visible camera
    pair lies left of it.  Short sequences / wide bands: no k-way order (the two-way split handles those)."""
    rng = np.random.default_rng(0)
    S, P = 1000, 20000
    start, ln = rng.integers(0, S - 40, P), rng.integers(5, 40, P)
    fr = np.arange(S)[:, None]
    m = torch.from_numpy((fr >= start[None]) & (fr < (start + ln)[None]))
    perm, fg = BA.find_camera_order(m)
    assert perm is not None and sorted(perm.tolist()) == list(range(S)) and perm[:2].tolist() == [0, 1]
    G = len(fg)
    roots = [p for p in range(G) if int(fg[p]) == p]                       # groups that start an independent block
    assert len(roots) >= 3
    fb = BA.envelope_blocks(fg, S, 6 * S + 2)
    assert fb.dtype == torch.int32 and len(fb) == (6 * S + 2 + 63) // 64 and int(fb[-1]) == 0      # intrinsics rows are dense
    mp = m[perm].float()
    covis = (mp @ mp.t()) > 0                                             # camera x camera in the new order
    first_cam = torch.tensor([int(torch.nonzero(covis[c])[0]) if bool(covis[c].any()) else c for c in range(S)])
    rows = torch.arange(6 * S)
    first_col = 6 * first_cam[rows // 6]
    for r in range(0, 6 * S, 64):
        assert int(fb[r // 64]) * 64 <= int(first_col[r:r + 64].min())
    # the envelope is far from dense: most block rows start well to the right
    assert float((fb[:-1] > 0).float().mean()) > 0.8
    sc = make_scene(200, 4000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
    assert BA.find_camera_order(T(sc.mask))[0] is None                     # band of 40 % of the frames: two-way split instead
    assert BA.find_camera_order(m[:60])[0] is None


def test_unknown_tile_order_is_refused(monkeypatch):
    """an unknown launch order used to be ignored silently (and an A/B variant relied on it)."""
    monkeypatch.setattr(BA, "TILE_ORDER", "no_such_order")
    sc = make_scene(20, 60, "SIMPLE_RADIAL", shared_camera=True, seed=3)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=3)
    with pytest.raises(ValueError, match="tile order"):
        BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL")


def test_environment_switches_are_inert_without_the_debug_gate(monkeypatch):
    monkeypatch.delenv("VGGSFM_AMD_DEBUG_HOOKS", raising=False)
    monkeypatch.setenv("VGGSFM_TILE_ORDER", "no_such_order")
    monkeypatch.setenv("VGGSFM_TILE_WGS", "0.01,0.01")
    sc = make_scene(20, 60, "SIMPLE_RADIAL", shared_camera=True, seed=3)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=3)
    BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL")


@pytest.mark.parametrize("S,N,cam,shared", [(17, 900, "SIMPLE_RADIAL", True), (40, 300, "SIMPLE_PINHOLE", True), (70, 400, "SIMPLE_RADIAL", False)])
def test_simple_work_list_covers_the_same_pairs(S, N, cam, shared, monkeypatch):
    """Round 6: below SIMPLE_WORKLIST_MAX_OBS observations (video windows) build_schur_tiles skips the density order of the
    tiles, the pattern grouping inside a tile and the cost model -- ~850 fewer operator calls per compile.  Same segments,
    same entries per tile (every co-observing camera-group pair of every point exactly once), plain sweep order inside a
    tile, one resident round."""
    sc = make_scene(S, N, cam, shared_camera=shared, seed=S)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=S)
    args = (T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), shared, cam)
    full, vf, _ = BA.compile_problem(*args)
    monkeypatch.setattr(BA, "SIMPLE_WORKLIST_MAX_OBS", 10 ** 9)
    monkeypatch.setattr(BA, "QUAD_SORT_WINDOW", 0)          # (what the checker reads: plain sweep order inside a tile)
    simp, vs, _ = BA.compile_problem(*args)
    vi, row_ptr, obs_cam, _, _ = OB.build_observations(pts0, ext0, sc.tracks, sc.mask)
    _check_views_and_work_list(simp, S, vi, row_ptr, obs_cam)
    assert simp.num_segments == full.num_segments and np.array_equal(simp.obs_slot.numpy(), full.obs_slot.numpy())
    key = lambda p: sorted(map(tuple, p.entries.numpy()[:, :3].tolist()))
    assert key(simp) == key(full)                           # the same (point, segment, segment) triples
    tiles = lambda p: sorted((int(a), int(b)) for a, b in p.tile_desc.numpy()[:, :2])
    assert tiles(simp) == tiles(full)


def test_group_adjacency_without_blas_matches_the_matrix_product():
    """ba._group_adjacency packs the points' camera-group sets into 62-bit words and combines the distinct sets pairwise (no GEMM:
    its first call costs a 160 ms library load); same (G, G) table as (V V^T) > 0, also beyond 62 groups and with a reduce hook."""
    torch.manual_seed(0)
    for S, P, dens in ((200, 5000, 0.25), (77, 300, 0.1), (1100, 3000, 0.01), (40, 10, 0.5), (1000, 20000, None)):
        if dens is None:                                                   # banded (video-like) visibility
            f = torch.arange(S)[:, None]
            c = torch.randint(0, S, (P,))[None]
            masks = (f - c).abs() < 20
        else:
            masks = torch.rand(S, P) < dens
        G = (S + 15) // 16
        pad = G * 16 - S
        m = torch.cat([masks, masks.new_zeros((pad, P))]) if pad else masks
        V = m.reshape(G, 16, -1).any(1).float()
        ref = (V @ V.t()) > 0
        assert torch.equal(BA._group_adjacency(masks, 16, None), ref)
        seen = []
        got = BA._group_adjacency(masks, 16, lambda t: (seen.append((t.dtype, tuple(t.shape))), t.fill_(1.0)))
        assert seen == [(torch.float32, (G, G))] and bool(got.all())


def test_split_exchange_partition_counts_from_the_c_abi():
    """vgg_ba_reduce_buffer(which = 7) -- host code, no GPU -- returns the number of elements of part A of the split exchange
    (include/vggsfm_amd.h, phases 7..12): the lower-triangle elements whose row and column belong to cameras of DIFFERENT
    16-camera groups.  Restated here element by element for 6 x 6 (shared / constant intrinsics), 7 x 7 and 8 x 8 blocks, with a
    partial last group; together with part B (same group, the shared border, the right-hand side) it is the whole payload."""
    L = _lib.lib()
    for C, model, rf, rk, shared in ((40, 1, 1, 1, False), (37, 0, 1, 0, False), (50, 1, 1, 1, True), (33, 0, 0, 0, True), (16, 1, 1, 1, False),
                                     (17, 1, 0, 1, False)):
        pb = _lib.BAProblem()
        pb.num_cams, pb.num_pts, pb.num_obs, pb.num_intr = C, 10, 100, 1 if shared else C
        pb.camera_model, pb.refine_focal, pb.refine_extra = model, rf, rk
        pb.num_chunks, pb.num_tile_batches, pb.num_segments, pb.num_tiles = 4, 1, 8, 3
        op = _lib.BAOptions()
        op.max_num_iterations = 5
        kd = rf + (1 if (rk and model == 1) else 0)
        n = 6 * C + kd * (1 if shared else C)
        fake = (ctypes.c_char * 64)()                               # (the carve-up only does address arithmetic)
        cnt, ptr = ctypes.c_size_t(), ctypes.POINTER(ctypes.c_double)()
        assert L.vgg_ba_reduce_buffer(ctypes.byref(pb), ctypes.byref(op), fake, 4, ctypes.byref(ptr), ctypes.byref(cnt)) == 0
        assert cnt.value == n * (n + 1) // 2 + n
        assert L.vgg_ba_reduce_buffer(ctypes.byref(pb), ctypes.byref(op), fake, 7, ctypes.byref(ptr), ctypes.byref(cnt)) == 0

        def group(x):                                               # camera group of a reduced index; -1: shared-intrinsics border
            if x < 6 * C:
                return (x // 6) // 16
            return -1 if (shared or kd == 0) else ((x - 6 * C) // kd) // 16
        idx = np.arange(n)
        g = np.array([group(int(x)) for x in idx])
        r, c = np.meshgrid(idx, idx, indexing="ij")
        lower = c <= r
        part_a = lower & (g[r] != g[c]) & (g[r] >= 0) & (g[c] >= 0)
        assert cnt.value == int(part_a.sum()), (C, model, rf, rk, shared)
