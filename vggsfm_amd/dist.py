"""Multi-GPU bundle adjustment: tracks sharded by 3D point, cameras replicated, one all-reduce of the
per-camera blocks / reduced camera system per LM iteration.

The reference has no distributed code at all; its serial chunking over the track axis
(vggsfm/utils/triangulation.py:712-758, triangulation_helpers.py:167-197) is the evidence that tracks are
independent.  Every residual touches exactly one point, so V_p, g_p, the Schur contribution and the
back-substitution of a point are rank-local; what couples the ranks is only the camera side:

  phase 0 LINEARIZE  local U_c = sum F^T F, g_c = sum F^T r, cost        -> all-reduce SUM  (C*(BD^2+BD+1) doubles)
  phase 1 SCHUR      local reduced system S, rhs of the rank's points     -> REDUCE-SCATTER + ALL-GATHER of the packed lower
                     local max |g_p|                                          triangle + rhs (n(n+1)/2 + n doubles: every rank
                                                                              sums 1/N of it, then the slices go round; on the
                                                                              point-to-point xGMI mesh that keeps all 7 links of
                                                                              a GPU busy); the local maxima ride in
                                                                              the gather (one slot per rank) -- no MAX collective
  phase 2 STEP       every rank factors S redundantly, back-substitutes its points,
                     local candidate cost / model change / step norm       -> all-reduce SUM  (4 doubles)
  phase 3 UPDATE     identical trust-region decision on every rank

Three collectives per LM iteration (round 2: four, the big one a plain all-reduce).  One process per GPU,
``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm).  The collectives run on the same stream as the
kernels, so the host never synchronises inside the loop.
"""
import ctypes

import torch

from . import _lib
from . import ba as BA
from .ba_options import LOSS_ID, BundleAdjustmentOptions


def partition_points(obs_per_point, world_size):
    """Contiguous point ranges balanced by observation count (prefix sum of the CSR row lengths).
    Returns a (world_size+1,) long tensor of boundaries, boundaries[0] = 0, boundaries[-1] = P."""
    counts = obs_per_point.to(torch.int64)
    P = counts.shape[0]
    csum = torch.cumsum(counts, 0)
    total = int(csum[-1].item()) if P > 0 else 0
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        idx = int(torch.searchsorted(csum, torch.tensor(target, dtype=csum.dtype, device=csum.device)).item())
        bounds.append(max(bounds[-1], min(idx, P)))
    bounds.append(P)
    return torch.tensor(bounds, dtype=torch.long)


def shard_slice(tracks, masks, points3d, rank, world_size):
    """The rank's contiguous slice of the track axis, balanced by observation count."""
    b = partition_points(masks.sum(0), world_size)
    lo, hi = int(b[rank]), int(b[rank + 1])
    return tracks[:, lo:hi], masks[:, lo:hi], points3d[lo:hi], (lo, hi)


def chunk_ranges(num_chunks, world_size):
    """Contiguous, near-equal ranges of the reference's track chunks for the ranks: (world_size + 1,) boundaries."""
    return [(num_chunks * r) // world_size for r in range(world_size + 1)]


def _gather_rows(local, counts, group=None):
    """all-gather of per-rank row blocks of different lengths (counts known on every rank) -> concatenation in rank order."""
    import torch.distributed as dist
    cap = max(counts) if counts else 0
    if cap == 0:
        return local
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in counts]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def _adopt_rank0_rng(device, group=None):
    """Every rank draws from rank 0's global CPU RNG state for the duration of one call (5 KB broadcast; through the device
    for RCCL, which moves device memory only).  Returns the rank's OWN state from before the call: the caller puts it back
    afterwards on every rank but the source, so that later per-rank randomness (sampling, shuffles) stays per-rank
    (ADVICE r4); rank 0 keeps the stream the draws advanced, as a single-rank run would.  A collective: every rank of the
    group must make the call."""
    import torch.distributed as dist
    own = torch.get_rng_state()
    on_device = dist.get_backend(group) == "nccl"
    buf = own.to(device) if on_device else own.clone()
    dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    torch.set_rng_state(buf.cpu())
    return own


def triangulate_tracks_sharded(extrinsics, tracks_normalized, rank, world_size, track_vis=None, track_score=None, gather=True,
                               group=None, sync_rng=True, **kw):
    """`triangulate_tracks` with the track axis shared by the ranks ("shard with no collective").  The unit is
    the reference's own chunk (triangulation.py:712-758: one randperm draw and one chunk-global indicator threshold per
    chunk), so every rank -- seeded alike -- draws the pairs of all chunks and triangulates a contiguous range of them: the
    concatenation over the ranks is bit for bit the single-rank result.  configs[2] has 25 chunks, configs[3] 147.
    gather=True: all-gather of points / inlier counts / masks (the only communication); False: the rank's own slice and
    its track range.
    sync_rng (default): the draws come from the global CPU RNG, which a rank may have consumed differently before this call
    (data loading, per-rank seeding) -- the gathered result would then silently differ from the single-rank one -- so every
    rank draws from rank 0's RNG state for this call and gets its own state back afterwards (only rank 0's
    stream is advanced by the draws); a COLLECTIVE -- every rank must pass the same sync_rng; False: the caller guarantees
    identical states (or has no process group: the lock-step tests)."""
    from .utils.triangulation import reference_chunks, triangulate_tracks
    own_rng = None
    if sync_rng and world_size > 1:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            own_rng = _adopt_rank0_rng(tracks_normalized.device, group)
    S, N = tracks_normalized.shape[0], tracks_normalized.shape[1]
    chunk_size, num_chunks = reference_chunks(S, N, kw.get("max_tri_points_num", 819200))
    b = chunk_ranges(num_chunks, world_size)
    try:
        pts, num, msk = triangulate_tracks(extrinsics, tracks_normalized, track_vis=track_vis, track_score=track_score,
                                           chunk_range=(b[rank], b[rank + 1]), **kw)
    finally:
        if own_rng is not None and rank != 0:
            torch.set_rng_state(own_rng)
    if not gather or world_size == 1:
        return pts, num, msk, (min(N, b[rank] * chunk_size), min(N, b[rank + 1] * chunk_size))
    counts = [min(N, b[r + 1] * chunk_size) - min(N, b[r] * chunk_size) for r in range(world_size)]
    return (_gather_rows(pts, counts, group), _gather_rows(num, counts, group),
            _gather_rows(msk.to(torch.uint8), counts, group).bool(), (0, N))


def filter_all_points3D_sharded(points3D, points2D, extrinsics, intrinsics, rank, world_size, extra_params=None, gather=True,
                                group=None, **kw):
    """`filter_all_points3D` on the rank's contiguous slice of the points (per-point work, any split gives the same
    masks); gather=True: all-gather of the masks."""
    from .utils.triangulation_helpers import filter_all_points3D
    P = points3D.shape[0]
    b = [(P * r) // world_size for r in range(world_size + 1)]
    lo, hi = b[rank], b[rank + 1]
    mask, det = filter_all_points3D(points3D[lo:hi], points2D[:, lo:hi], extrinsics, intrinsics, extra_params=extra_params, **kw)
    if not gather or world_size == 1:
        return (mask, det), (lo, hi)
    counts = [b[r + 1] - b[r] for r in range(world_size)]
    mask = _gather_rows(mask.to(torch.uint8), counts, group).bool()
    if det is not None:
        det = _gather_rows(det.t().contiguous().to(torch.uint8), counts, group).t().bool()
    return (mask, det), (0, P)


SPLIT_EXCHANGE = False       # default of ShardedBA(split_exchange=None); see ShardedBA.__init__


class Collectives:
    """The three exchanges of one LM iteration over ``torch.distributed``.

    `system_scatter(padded, mine, gathered)`: SUM of the packed reduced system over the ranks as a reduce-scatter + an
    all-gather on the solver's OWN buffers (``vgg_ba_reduce_buffer`` 4 / 5 / 6; include/vggsfm_amd.h, phase 6): `padded` =
    the packed system zero-padded to W equal slices by the pack kernel, `mine` = this rank's reduced slice with its local
    gradient maximum behind it, `gathered` = the W slices, which the unpack kernel reads in place -- no staging copy on
    either side (round 3 copied the payload twice, 41 MB each at configs[3]) and no MAX collective.  RCCL on the GPU box;
    gloo implements both calls too, which is how the CPU tests run this very code.
    `system_(packed, local_max)`: the same on caller-owned tensors through internal staging buffers (tests);
    `plain_all_reduce=True` keeps the round-2 form -- two all-reduces -- for A/B measurements."""

    def __init__(self, world_size, group=None, plain_all_reduce=False, host_staging=None):
        import torch.distributed as dist
        self.dist, self.world, self.group = dist, world_size, group
        self.scatter = not plain_all_reduce
        self._bufs = {}
        # gloo moves host memory: device tensors are staged through the host (rehearsals of the N > 1 flow on a box with
        # fewer GPUs than ranks -- bench.py with VGGSFM_BENCH_BACKEND=gloo; never the measured path, which is RCCL)
        self.host_staging = (dist.is_initialized() and dist.get_backend(group) == "gloo") if host_staging is None else host_staging

    def _staged(self, t):
        return self.host_staging and t.is_cuda

    def sum_(self, t):
        if self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def system_scatter(self, padded, mine, gathered):
        dist = self.dist
        if self._staged(padded):
            hp, hm, hg = padded.cpu(), mine.cpu(), torch.empty(gathered.shape, dtype=gathered.dtype)
            dist.reduce_scatter_tensor(hm[:-1], hp, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(hg, hm, group=self.group)
            mine.copy_(hm)
            gathered.copy_(hg)
            return
        dist.reduce_scatter_tensor(mine[:-1], padded, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_gather_into_tensor(gathered, mine, group=self.group)

    def system_scatter_begin(self, padded, mine, gathered, rides=1):
        """`system_scatter` without waiting for it: the two collectives are enqueued on the communicator's own stream behind what the
        current stream holds at this point (torch's async_op), so that kernels launched afterwards run beside them;
        `system_scatter_end(handle)` makes the current stream wait.  `rides`: elements behind the rank's slice that travel in the
        gather only (1 = the gradient maximum; 0 for a part without one).  Host-staged transports run it synchronously."""
        dist = self.dist
        own = mine[:mine.numel() - rides] if rides else mine
        if self._staged(padded):
            hp, hm, hg = padded.cpu(), mine.cpu(), torch.empty(gathered.shape, dtype=gathered.dtype)
            dist.reduce_scatter_tensor(hm[:hm.numel() - rides] if rides else hm, hp, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(hg, hm, group=self.group)
            mine.copy_(hm)
            gathered.copy_(hg)
            return None
        if not (dist.is_initialized() and dist.get_backend(self.group) == "nccl"):
            # (only RCCL orders the two on a stream: with another transport the gather could read the slice before the
            #  reduce-scatter has written it)
            dist.reduce_scatter_tensor(own, padded, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(gathered, mine, group=self.group)
            return None
        w1 = dist.reduce_scatter_tensor(own, padded, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        w2 = dist.all_gather_into_tensor(gathered, mine, group=self.group, async_op=True)
        return (w1, w2)

    @staticmethod
    def system_scatter_end(handle):
        if handle is not None:
            handle[0].wait()
            handle[1].wait()

    def system_(self, packed, local_max):
        dist, W = self.dist, self.world
        if not self.scatter:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(local_max, op=dist.ReduceOp.MAX, group=self.group)
            return
        M = packed.numel()
        chunk = -(-M // W)
        key = (M, packed.device)
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros(W * chunk, dtype=packed.dtype, device=packed.device),
                               torch.empty(chunk + 1, dtype=packed.dtype, device=packed.device),
                               torch.empty(W * (chunk + 1), dtype=packed.dtype, device=packed.device))
        padded, mine, gathered = self._bufs[key]
        padded[:M].copy_(packed)                                  # (the tail stays zero)
        mine[chunk:].copy_(local_max[:1])
        self.system_scatter(padded, mine, gathered)
        g = gathered.view(W, chunk + 1)
        packed.copy_(g[:, :chunk].reshape(-1)[:M])
        local_max[:1].copy_(g[:, chunk].max().reshape(1))


class _FunctionCollectives:
    """Adapter for a caller-supplied ``all_reduce(tensor, "sum" | "max")`` (tests, lock-step emulation)."""

    def __init__(self, fn):
        self.fn = fn

    def sum_(self, t):
        self.fn(t, "sum")

    def system_(self, packed, local_max):
        self.fn(packed, "sum")
        self.fn(local_max, "max")


class ShardedBA:
    """LM loop of one rank's DeviceProblem with the collectives interleaved (also used with world_size 1)."""

    def __init__(self, problem, options=None, rank=0, world_size=1, all_reduce=None, collectives=None, split_exchange=None):
        """split_exchange (round 6; None = `SPLIT_EXCHANGE`): exchange the reduced system in two parts -- the off-diagonal tile
        launch's share travels on the communicator's stream while the diagonal launch computes the rest (phases 7..11 of
        include/vggsfm_amd.h).  Needs `Collectives` (reduce-scatter + all-gather) and a problem with one tile batch and separate
        tile launches; otherwise the one-piece exchange is used."""
        self.L = _lib.lib()
        self.problem = problem
        self.options = options or BundleAdjustmentOptions()
        problem.refine_focal = self.options.refine_focal_length
        problem.refine_extra = self.options.refine_extra_params
        problem.loss = LOSS_ID[self.options.loss_function_type]
        problem.loss_scale = self.options.loss_function_scale
        self.rank, self.world = rank, world_size
        self.cp = problem.c_struct()
        self.co = BA._c_options(self.options, overlap=(world_size == 1))
        nbytes = int(self.L.vgg_ba_workspace_bytes(ctypes.byref(self.cp), ctypes.byref(self.co)))
        self.nbytes = nbytes
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=problem.pts.device)
        self.bufs = []
        for which in range(7):
            p = ctypes.POINTER(ctypes.c_double)()
            cnt = ctypes.c_size_t()
            _lib.check(self.L.vgg_ba_reduce_buffer(ctypes.byref(self.cp), ctypes.byref(self.co), _lib.ptr(self.ws), which,
                                                   ctypes.byref(p), ctypes.byref(cnt)), "vgg_ba_reduce_buffer")
            off = ctypes.addressof(p.contents) - self.ws.data_ptr()
            self.bufs.append(self.ws[off:off + 8 * cnt.value].view(torch.float64))
        if collectives is None and world_size > 1:
            collectives = _FunctionCollectives(all_reduce) if all_reduce is not None else Collectives(world_size)
        # collectives handed in explicitly are used at world size 1 as well: a one-rank RCCL communicator runs the whole
        # exchange sequence on the solver's buffers (tests/test_gpu_dist_rccl.py::test_rccl_one_rank_...)
        self.coll = collectives
        # reduce-scatter / all-gather views of the solver's own buffers (phase 4 pads, phase 6 reads the gathered slices)
        M = self.bufs[4].numel()
        chunk = -(-M // max(world_size, 1))
        off4 = self.bufs[4].data_ptr() - self.ws.data_ptr()
        self._padded = self.ws[off4:off4 + 8 * world_size * chunk].view(torch.float64)
        self._mine = self.bufs[5][:chunk + 1]
        self._gathered = self.bufs[6][:world_size * (chunk + 1)]
        self._in_place = isinstance(collectives, Collectives) and collectives.scatter
        # split exchange: [A | B] regions of the same three buffers (include/vggsfm_amd.h, phases 7..11)
        want_split = SPLIT_EXCHANGE if split_exchange is None else bool(split_exchange)
        self._split = False
        if want_split and (self._in_place or split_exchange == "emulated"):
            rc = self.L.vgg_ba_phase(ctypes.byref(self.cp), ctypes.byref(self.co), _lib.ptr(self.ws), 12, _lib.stream_ptr())
            if rc == 0:
                p = ctypes.POINTER(ctypes.c_double)()
                cnt = ctypes.c_size_t()
                _lib.check(self.L.vgg_ba_reduce_buffer(ctypes.byref(self.cp), ctypes.byref(self.co), _lib.ptr(self.ws), 7,
                                                       ctypes.byref(p), ctypes.byref(cnt)), "vgg_ba_reduce_buffer")
                W = max(world_size, 1)
                a = int(cnt.value)
                ca, cb = -(-a // W), -(-(M - a) // W)
                b4 = self.ws[off4:off4 + 8 * W * (ca + cb)].view(torch.float64)
                self._padded_a, self._padded_b = b4[:W * ca], b4[W * ca:]
                self._mine_a, self._mine_b = self.bufs[5][:ca], self.bufs[5][ca:ca + cb + 1]
                self._gathered_a, self._gathered_b = self.bufs[6][:W * ca], self.bufs[6][W * ca:W * ca + W * (cb + 1)]
                self._split = True

    def begin(self):
        _lib.check(self.L.vgg_ba_begin(ctypes.byref(self.cp), ctypes.byref(self.co), _lib.ptr(self.ws),
                                       ctypes.c_size_t(self.nbytes), self.rank, self.world, _lib.stream_ptr()), "vgg_ba_begin")

    def _phase(self, i):
        _lib.check(self.L.vgg_ba_phase(ctypes.byref(self.cp), ctypes.byref(self.co), _lib.ptr(self.ws), i,
                                       _lib.stream_ptr()), "vgg_ba_phase")

    def iteration(self):
        co = self.coll
        self._phase(0)
        if co:
            co.sum_(self.bufs[0])
        if co and self._split:
            # the off-diagonal tiles' share of the system is on its way while the diagonal tile launch runs
            self._phase(7)
            self._phase(8)
            ha = co.system_scatter_begin(self._padded_a, self._mine_a, self._gathered_a, rides=0)
            self._phase(9)
            self._phase(10)
            hb = co.system_scatter_begin(self._padded_b, self._mine_b, self._gathered_b, rides=1)
            co.system_scatter_end(ha)
            co.system_scatter_end(hb)
            self._phase(11)
            self._phase(2)
            co.sum_(self.bufs[3])
            self._phase(3)
            return
        self._phase(1)
        if co:
            self._phase(4)                      # lower triangle + rhs -> packed buffer (half the payload), padded to W slices
            if self._in_place:
                co.system_scatter(self._padded, self._mine, self._gathered)
                self._phase(6)                  # S | rhs and the gradient maximum straight from the gathered slices
            else:
                co.system_(self.bufs[4], self.bufs[2])
                self._phase(5)
        self._phase(2)
        if co:
            co.sum_(self.bufs[3])
        self._phase(3)

    def finish(self, log_cap=0):
        summ = _lib.BASummary()
        log = (_lib.BAIteration * max(log_cap, 1))()
        _lib.check(self.L.vgg_ba_finish(ctypes.byref(self.cp), ctypes.byref(self.co), _lib.ptr(self.ws), ctypes.byref(summ),
                                        log if log_cap else None, log_cap, _lib.stream_ptr()), "vgg_ba_finish")
        return BA._summary_dict(summ, log, summ.num_log if log_cap else 0)

    POLL = 8

    def done(self):
        """Has the device-side control flow ended the solve?  (4-byte copy + stream synchronisation.)"""
        flag = ctypes.c_int32(0)
        _lib.check(self.L.vgg_ba_poll_done(ctypes.byref(self.cp), ctypes.byref(self.co), _lib.ptr(self.ws), ctypes.byref(flag),
                                           _lib.stream_ptr()), "vgg_ba_poll_done")
        return bool(flag.value)

    def solve(self):
        """Whole solve.  Termination is decided on the device and identically on every rank (the decision inputs are
        all-reduced); iterations enqueued after it are no-ops but still cost their launches and collectives, so the host
        reads the `done` flag every POLL iterations -- every rank at the same iteration, hence the same answer -- and
        stops enqueuing (what ``vgg_ba_solve`` does for the single-GPU path)."""
        self.begin()
        cap = self.options.solver_options.max_num_iterations
        for it in range(cap + 1):
            self.iteration()
            if it % self.POLL == self.POLL - 1 and it < cap and self.done():
                break
        return self.finish(cap + 2)
