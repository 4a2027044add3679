"""Progress argument of the single-launch factorisation (vggsfm_amd/csrc/chol.hip, chol_dataflow_kernel), replayed on the
host: which workgroup waits for which flag and who raises it, for the role assignment of the CHAINED form (the workgroup of
the sub-diagonal tile (c+1, c) also finishes the diagonal tile (c+1, c+1)) and for the plain one.  The kernel's claim is
that a workgroup only ever waits for workgroups EARLIER in launch order (dispatch is in order, so the earliest unfinished
workgroup is always resident and can run to completion) -- with ONE exception since round 6: the merged workgroup of tile
(r, r - 1) waits for `dready[r]`, raised by the diagonal tile's own workgroup (r, r), which applies that tile's earlier
updates and sits a few positions LATER in the launch.  That workgroup itself waits only for earlier ones, and everything
between the two in launch order can finish without it, so a bounded window of resident workgroups still drains: checked
below by simulating in-order dispatch into windows of 2 .. 256 slots.  Checked for dense matrices, the camera split and
k-way / banded envelopes: every awaited flag is raised by exactly one workgroup, earlier except for dready, no flag is raised
twice, and the depth order of the left-looking updates is a permutation that equals the ascending one when the matrix is dense.
(This is a model of the kernel's control flow, kept next to the envelope builders it is exercised with; the numerics are
covered by tests/test_gpu_ba.py::test_cholesky_*.)"""
import numpy as np
import pytest
import torch

from vggsfm_amd import ba as BA

DFB = 64


def first_of_factory(nbk, first_blk=None, split=(0, 0)):
    sa, sb = split
    if not (sa >= DFB and sb >= DFB and sa % DFB == 0 and sa + sb <= nbk * DFB):
        sa = sb = 0

    def first_of(br):
        if first_blk is not None:
            return int(first_blk[br]) if br < nbk else 0
        return sa // DFB if (sb > 0 and br < nbk and DFB * br >= sa and DFB * (br + 1) <= sa + sb) else 0
    return first_of


def depth_order(first_of, kstart, c, kend=None):
    depth = []
    for k in range(c):
        f = first_of(k)
        depth.append(1 + max([depth[j] for j in range(f, k)], default=0))
    ks = list(range(kstart, c if kend is None else kend))
    return sorted(ks, key=lambda k: (depth[k], k))


def tile_map(nbk, first_of, chain):
    """df_tile_map_kernel restated: the launch order used with a row envelope -- columns by dependency depth (the merged
    workgroup of tile (c + 1, c) also takes the updates of row c + 1: kept from round 5, harmless), then by index; tiles
    outside the envelope are left out (round 6: the diagonal tiles a merged workgroup finishes are IN -- their own workgroup
    applies their earlier updates).  -> list of (c, r)"""
    def chained(x):
        return chain and 1 <= x < nbk and first_of(x) <= x - 1
    dep = []
    for c in range(nbk):
        lo = first_of(c)
        if chained(c + 1):
            lo = min(lo, first_of(c + 1))
        dep.append(1 + max([dep[k] for k in range(lo, c)], default=0))
    cols = sorted(range(nbk), key=lambda c: (dep[c], c))
    return [(c, r) for c in cols for r in range(c, nbk + 1) if first_of(r) <= c]


def schedule(nbk, first_of, chain, order=None):
    """-> list of workgroups in launch order: dict(tile, waits [flag names in program order], raises [flag names])"""
    def chained(x):
        return chain and 1 <= x < nbk and first_of(x) <= x - 1
    wgs = []
    for c, r in (order if order is not None else [(c, r) for c in range(nbk) for r in range(c, nbk + 1)]):
        if True:
            wg = dict(tile=(r, c), waits=[], raises=[])
            wgs.append(wg)
            if c < first_of(r):
                continue                                             # structurally zero tile
            diag = r == c
            prep = diag and chained(c)                               # finished by the workgroup of tile (c, c - 1): early updates here
            merged = (not diag) and r == c + 1 and chained(r)
            kfirst = max(first_of(r), first_of(c))
            kstart, kend = kfirst, (c - 1 if prep else c)
            n_upd = max(kend - kstart, 0)
            order = depth_order(first_of, kstart, c, kend) if (chain and n_upd >= 2) else list(range(kstart, kend))
            assert sorted(order) == list(range(kstart, kend))
            for k in order:
                wg["waits"].append(("ready", r, k))
                if not diag and k >= kfirst:
                    wg["waits"].append(("ready", c, k))
            if prep:
                wg["raises"].append(("dready", c))
                continue
            if diag:
                wg["raises"].append(("tready", c))
                continue
            wg["waits"].append(("tready", c))
            if merged:
                wg["waits"].append(("dready", r))                    # (in front of the last slab: before anything is raised)
            wg["raises"].append(("ready", r, c))
            if merged:
                wg["raises"].append(("tready", r))
    return wgs


def check(nbk, first_of, chain, order=None):
    wgs = schedule(nbk, first_of, chain, order)
    raised_by = {}
    for i, wg in enumerate(wgs):
        for f in wg["raises"]:
            assert f not in raised_by, f"flag {f} raised twice"
            raised_by[f] = i
    for i, wg in enumerate(wgs):
        for f in wg["waits"]:
            assert f in raised_by, f"workgroup {wg['tile']} waits for {f}, which nobody raises"
            if f[0] != "dready":
                assert raised_by[f] < i, f"workgroup {wg['tile']} (launch position {i}) waits for a later one ({raised_by[f]})"
    # every block column that has tiles below its diagonal gets its T published
    for c in range(nbk):
        assert ("tready", c) in raised_by
    # in-order dispatch into a bounded window of resident workgroups drains (the dready waits point FORWARD in the launch: one
    # merged workgroup per pivot chain can sit waiting for a later one, so the window must hold the chains + one that works --
    # 256 CUs against the handful of chains of a k-way camera order)
    chains = sum(1 for c in range(nbk) if not (chain and c >= 1 and first_of(c) <= c - 1))
    for window in (chains + 1, chains + 8, 256):
        up, done, resident, nxt = set(), 0, [], 0
        while done < len(wgs):
            while len(resident) < window and nxt < len(wgs):
                resident.append(nxt)
                nxt += 1
            ready = [i for i in resident if all(f in up for f in wgs[i]["waits"])]
            assert ready, f"window {window}: resident workgroups {[wgs[i]['tile'] for i in resident]} all wait"
            for i in ready:
                up.update(wgs[i]["raises"])
                resident.remove(i)
                done += 1
    return wgs


@pytest.mark.parametrize("chain", [False, True])
@pytest.mark.parametrize("n,split", [(130, (0, 0)), (1202, (0, 0)), (1202, (384, 288)), (700, (128, 192)), (3200, (1024, 960)),
                                     (450, (64, 300))])
def test_dense_and_split_schedules_wait_only_for_earlier_workgroups(n, split, chain):
    nbk = (n + DFB - 1) // DFB
    check(nbk, first_of_factory(nbk, None, split), chain)


@pytest.mark.parametrize("chain", [False, True])
def test_envelope_schedules_wait_only_for_earlier_workgroups(chain):
    rng = np.random.default_rng(0)
    for trial in range(40):
        nbk = int(rng.integers(2, 40))
        # a random row envelope: first_blk[r] <= r, several decoupled leading blocks, dense last rows
        first = np.zeros(nbk, np.int64)
        start = 0
        for r in range(nbk):
            if rng.random() < 0.2:
                start = r                                             # a new decoupled block starts here
            first[r] = rng.integers(start, r + 1) if rng.random() < 0.7 else start
        first[-1] = 0
        fo = first_of_factory(nbk, first)
        plain = check(nbk, fo, chain)
        # the depth-ordered launch map of the enveloped launch: the same workgroups (minus the idle ones), still a topological order
        mapped = check(nbk, fo, chain, tile_map(nbk, fo, chain))
        busy = sorted(w["tile"] for w in plain if w["waits"] or w["raises"])
        assert sorted(w["tile"] for w in mapped) == busy


def test_video_envelope_from_the_camera_order():
    """the k-way camera order of a sliding-window visibility (ba.find_camera_order + envelope_blocks), as BASELINE configs[4]"""
    rng = np.random.default_rng(0)
    S, P = 1000, 20000
    start, ln = rng.integers(0, S - 40, P), rng.integers(5, 40, P)
    fr = np.arange(S)[:, None]
    masks = torch.from_numpy((fr >= start[None]) & (fr < (start + ln)[None]))
    perm, first_group = BA.find_camera_order(masks)
    assert perm is not None
    n = 6 * S + 2
    first_blk = BA.envelope_blocks(first_group, S, n).numpy()
    nbk = (n + DFB - 1) // DFB
    assert (first_blk[1:nbk - 1] > 0).any()
    for chain in (False, True):
        fo = first_of_factory(nbk, first_blk)
        check(nbk, fo, chain)
        order = tile_map(nbk, fo, chain)
        check(nbk, fo, chain, order)
    # every interior run starts on a 64-column block: its pivot chain is its own (block row r with first_blk[r] == r)
    starts = [r for r in range(nbk) if first_blk[r] == r]
    assert len(starts) >= 4, starts
    # ... and the depth order interleaves the chains: the launch begins with the first column of EVERY run
    cols_in_order = list(dict.fromkeys(c for c, r in order))
    assert sorted(cols_in_order[:len(starts)]) == starts


def test_depth_order_is_ascending_for_a_dense_matrix_and_puts_the_shorter_chain_first():
    nbk = 19
    dense = first_of_factory(nbk)
    assert depth_order(dense, 0, 12) == list(range(12))
    split = first_of_factory(nbk, None, (384, 288))                  # A = block columns 0..5, B = 6..9 (9 is partly "rest")
    order = depth_order(split, 0, 10)
    # the columns of both chains interleave by depth; the LAST ones are those of the longer chain (A: depth 5, 6)
    assert order[-2:] == [4, 5] and set(order) == set(range(10))
    assert order.index(6) < order.index(1)                           # B's first column (depth 1) before A's second (depth 2)
