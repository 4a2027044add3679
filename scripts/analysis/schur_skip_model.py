"""Cost model of the Schur tile kernels on the visibility of bench workload c3 (numpy, no GPU): executed MFMAs and staged
bytes of (a) today's kernel, (b) sub-tile skipping by per-batch presence masks, (c) 2x2 super-tiles (192 x 192, 8
wavefronts) with skipping.  Used to decide what to build."""
import numpy as np

S, N = 200, 100000
G = 16
rng = np.random.Generator(np.random.PCG64(1000))
_ = rng.uniform(-1.0, 1.0, size=(N, 3))
lo, hi = max(3, int(np.ceil(0.1 * S))), int(np.ceil(0.4 * S))
length = rng.integers(lo, hi + 1, size=N)
start = (rng.uniform(0.0, 1.0, size=N) * (S - length + 1)).astype(np.int64)
end = start + length                       # exclusive
ng = (S + G - 1) // G


def slot_mask(g):
    """(N,) 16-bit presence of every point in camera group g"""
    a = np.clip(start - g * G, 0, G)
    b = np.clip(end - g * G, 0, G)
    m = np.where(b > a, ((1 << b) - 1) & ~((1 << a) - 1), 0)
    return m.astype(np.int64)


def rb_mask(m16, bd=6):
    """16-slot mask -> mask of the 16-row blocks (R = 16 bd rows) that hold a present camera"""
    nt = G * bd // 16
    out = np.zeros_like(m16)
    for rb in range(nt):
        s0, s1 = (16 * rb) // bd, (16 * rb + 15) // bd
        bits = ((1 << (s1 + 1)) - 1) & ~((1 << s0) - 1)
        out |= ((m16 & bits) != 0).astype(np.int64) << rb
    return out


masks = [slot_mask(g) for g in range(ng)]
rbm = [rb_mask(m) for m in masks]
popc = np.array([bin(i).count("1") for i in range(64)])


def batches(arr, pad=0):
    n = arr.shape[0]
    k = (-n) % 4
    if k:
        arr = np.concatenate([arr, np.full(k, pad, arr.dtype)])
    return arr.reshape(-1, 4)


def union4(arr):
    b = batches(arr)
    return b[:, 0] | b[:, 1] | b[:, 2] | b[:, 3]


def strided_count(m, nwaves_side, nt=6):
    """per wave-row r: number of active blocks among {r, r + nwaves_side, ...}"""
    out = []
    for r in range(nwaves_side):
        bits = sum(1 << b for b in range(r, nt, nwaves_side))
        out.append(popc[m & bits])
    return np.stack(out, 0)


order_key = start * 1000 + length
tot = dict(entries_off=0, entries_diag=0, base_off=0, skip_off=0, skip_off_sorted=0, skip_off_ideal=0, base_diag=0, useful=0)
for gi in range(ng):
    for gj in range(gi, ng):
        sel = np.nonzero((masks[gi] != 0) & (masks[gj] != 0))[0]
        if sel.size == 0:
            continue
        if gi == gj:
            tot["entries_diag"] += sel.size
            tot["base_diag"] += batches(sel).shape[0] * 6 * 3          # PER = 6 sub-tiles per wave
            continue
        tot["entries_off"] += sel.size
        nb = batches(sel).shape[0]
        tot["base_off"] += nb * 27
        for name, idx in (("skip_off", sel), ("skip_off_sorted", sel[np.argsort(order_key[sel], kind="stable")])):
            ra, cb = union4(rbm[gi][idx]), union4(rbm[gj][idx])
            wr, wc = strided_count(ra, 2), strided_count(cb, 2)
            tot[name] += int((wr.max(0) * wc.max(0)).sum()) * 3
        ra, cb = rbm[gi][sel], rbm[gj][sel]
        tot["skip_off_ideal"] += float((popc[ra] * popc[cb]).sum()) * 3 / 4 / 4   # per-entry masks, perfect balance over 4 waves
print({k: (int(v) if isinstance(v, (int, np.integer)) else round(v)) for k, v in tot.items()})
print("off-diagonal MFMA slots per wave: base %.3g, skip %.3g (%.2f), skip+sorted %.3g (%.2f), ideal %.3g (%.2f)" % (
    tot["base_off"], tot["skip_off"], tot["skip_off"] / tot["base_off"], tot["skip_off_sorted"],
    tot["skip_off_sorted"] / tot["base_off"], tot["skip_off_ideal"], tot["skip_off_ideal"] / tot["base_off"]))
seg_bytes = 16 * 18 * 8
print("staged bytes: off %.2f GB, diag %.2f GB" % (tot["entries_off"] * 2 * seg_bytes / 1e9, tot["entries_diag"] * seg_bytes / 1e9))

# ---- 2 x 2 super-tiles: super group = 32 cameras, 192 x 192, 8 wavefronts as 2 (rows) x 4 (cols) or 4 x 2 ----
SG = 2 * G
nsg = (S + SG - 1) // SG
smask = []
for sg in range(nsg):
    m = rbm[2 * sg].copy()
    if 2 * sg + 1 < ng:
        m |= rbm[2 * sg + 1] << 6
    smask.append(m)
popc12 = np.array([bin(i).count("1") for i in range(4096)])


def strided12(m, k):
    out = []
    for r in range(k):
        bits = sum(1 << b for b in range(r, 12, k))
        out.append(popc12[m & bits])
    return np.stack(out, 0)


for shape in ((2, 4), (4, 2)):
    t2 = dict(entries=0, segs=0, slots=0, diag_entries=0, diag_slots=0, ideal=0.0)
    for si in range(nsg):
        for sj in range(si, nsg):
            sel = np.nonzero((smask[si] != 0) & (smask[sj] != 0))[0]
            if sel.size == 0:
                continue
            idx = sel[np.argsort(order_key[sel], kind="stable")]
            ra, cb = union4(smask[si][idx]), union4(smask[sj][idx])
            if si == sj:
                # lower triangle of 12 x 12 sub-tiles dealt to 8 waves: model = active pairs (rb >= cb) / 8, rounded up
                t2["diag_entries"] += sel.size
                act = popc12[ra]
                t2["diag_slots"] += int(np.ceil(act * (act + 1) / 2 / 8).sum()) * 3
                nseg = popc12[(smask[si][sel] & 63 != 0).astype(int)] + (smask[si][sel] >> 6 != 0)
                t2["segs"] += int(nseg.sum())
                continue
            t2["entries"] += sel.size
            wr, wc = strided12(ra, shape[0]), strided12(cb, shape[1])
            t2["slots"] += int((wr.max(0) * wc.max(0)).sum()) * 3
            t2["ideal"] += float((popc12[smask[si][sel]] * popc12[smask[sj][sel]]).sum()) * 3 / 4 / 8
            nseg = (smask[si][sel] & 63 != 0).astype(int) + (smask[si][sel] >> 6 != 0) + (smask[sj][sel] & 63 != 0) + (smask[sj][sel] >> 6 != 0)
            t2["segs"] += int(nseg.sum())
    print("super 2x2, waves %s: off entries %d, MFMA slots per wave %.3g (+ diag %.3g), ideal %.3g; staged %.2f GB (non-empty segments only)" % (
        shape, t2["entries"], t2["slots"], t2["diag_slots"], t2["ideal"], t2["segs"] * seg_bytes / 1e9))
