#!/usr/bin/env python3
"""Residency of the two PRODUCT blend kernels (GSR_CENSUS build: same registers / LDS as libgsr_hip.so, three scalar stamps per wave).
usage: make -C 3dgs-native_amd/csrc census; GSR_LIB=$PWD/3dgs-native_amd/libgsr_hip_census.so python tools/residency.py [C3]
Prints, per kernel: span, waves per SIMD averaged over the span, the peak per CU / per SIMD, the mean residency curve along the
span, wave-life percentiles and how unevenly the CUs finish."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = gsr.scenes.CONFIGS[name]
sc = gsr.scenes.synthetic_scene(cfg["n"], cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
W, H = cfg["width"], cfg["height"]
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()
bg = np.zeros(3, np.float32)
P = dict(means3D=t(sc["means"]), opacity=t(sc["opacities"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
shs = t(sc["shs"])
kw = dict(background=bg, **P, viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
          image_height=H, image_width=W, sh=shs, degree=3, campos=cam["camera_center"])
dpix = t(np.random.default_rng(99).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3))
L = gsr._lib.lib()
tiles = ((W + 15) // 16) * ((H + 15) // 16)
nw = {"fwd": tiles * 4, "bwd": tiles * 8}
fn = {"fwd": L.gsr_debug_fwd_census, "bwd": L.gsr_debug_bwd_census}


def step():
    img, depth, buf = gsr.render_gaussians(**kw)
    gsr.backward(background=bg, dL_dpixels=dpix, shs=shs, **P, viewmatrix=kw["viewmatrix"], projmatrix=kw["projmatrix"], tan_fovx=kw["tan_fovx"],
                 tan_fovy=kw["tan_fovy"], image_height=H, image_width=W, campos=kw["campos"], radii=buf["radii"], means2D=buf["points_xy_image"],
                 conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"],
                 binning_buffer={"point_list": buf["point_list"]}, img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]})
    return buf


for _ in range(3):
    buf = step()
torch.cuda.synchronize()
for k in fn:
    assert fn[k](None, nw[k], 1) == 0
buf = step()
torch.cuda.synchronize()
ranges = buf["ranges"].cpu().numpy().reshape(-1, 2)
list_len = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)


def curve(keys, r0, r1, grid):
    peaks, curves = [], []
    for k in np.unique(keys):
        m = keys == k
        ev = np.concatenate([np.stack([r0[m], np.ones(m.sum(), np.int64)], 1), np.stack([r1[m], -np.ones(m.sum(), np.int64)], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        conc = np.cumsum(ev[:, 1])
        peaks.append(conc.max())
        curves.append([conc[max(0, np.searchsorted(ev[:, 0], g, side="right") - 1)] for g in grid])
    return np.array(peaks), np.array(curves, np.float64)


for k in ("fwd", "bwd"):
    arr = np.zeros((nw[k], 4), np.uint64)
    assert fn[k](arr.ctypes.data_as(C.c_void_p), nw[k], 0) == 0
    ran = arr[:, 1] > 0
    hw = (arr[ran, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((arr[ran, 0] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]
    cu_key = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    simd_key = (cu_key << 2) | ((hw >> 4) & 3)
    r0, r1 = arr[ran, 1].astype(np.int64), arr[ran, 2].astype(np.int64)
    t0, t1 = r0.min(), r1.max()
    span = float(t1 - t0)
    n_simd = np.unique(simd_key).size
    grid = t0 + (np.linspace(0.025, 0.975, 20) * span).astype(np.int64)
    print(f"== blend_{k} ({name}): {int(ran.sum())} waves ran of {nw[k]}; span {span / 100:.1f} us; {np.unique(cu_key).size} CUs / {n_simd} SIMDs seen")
    print(f"   mean resident waves per SIMD over the span: {(r1 - r0).sum() / span / n_simd:.2f}")
    pk, cv = curve(cu_key, r0, r1, grid)
    print(f"   peak resident waves per CU: min {pk.min()} median {int(np.median(pk))} max {pk.max()}  (32 = 8 per SIMD)")
    print("   mean resident waves per CU at 5 % steps of the span: " + " ".join(f"{v:.1f}" for v in cv.mean(axis=0)))
    pk, _ = curve(simd_key, r0, r1, grid[:1])
    print(f"   peak resident waves per SIMD: min {pk.min()} median {int(np.median(pk))} max {pk.max()}")
    life = (r1 - r0) / 100.0
    print(f"   wave life (us): p10 {np.percentile(life, 10):.1f} p50 {np.percentile(life, 50):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f}")
    # when does each CU run dry?
    last = np.array([r1[cu_key == c].max() for c in np.unique(cu_key)], np.float64)
    print(f"   CU finish time as a fraction of the span: p10 {np.percentile((last - t0) / span, 10):.2f} p50 {np.percentile((last - t0) / span, 50):.2f} p90 {np.percentile((last - t0) / span, 90):.2f}")
    start_frac = (r0 - t0) / span
    print(f"   wave start as a fraction of the span: p50 {np.percentile(start_frac, 50):.2f} p90 {np.percentile(start_frac, 90):.2f} p99 {np.percentile(start_frac, 99):.2f} max {start_frac.max():.2f}")
    # per XCD: when does it run dry, how many waves did it get
    xk = cu_key >> 12
    print("   per XCD (waves, finish as a fraction of the span): " + "  ".join(f"{x}: {int((xk == x).sum())} {(r1[xk == x].max() - t0) / span:.2f}" for x in np.unique(xk)))
    # life against the tile's list length (is the tail made of deep tiles?)
    per_tile = 4 if k == "fwd" else 8
    wave_tile = np.flatnonzero(ran) // per_tile
    ll = list_len[wave_tile]
    late = r1 > t0 + 0.85 * span
    print(f"   list length of the tile: all waves mean {ll.mean():.0f}; waves still alive after 85 % of the span: {int(late.sum())}, mean list {ll[late].mean() if late.any() else 0:.0f}, mean life {life[late].mean() if late.any() else 0:.1f} us, mean start {start_frac[late].mean() if late.any() else 0:.2f}")
    print(f"   correlation(life, list length) = {np.corrcoef(life, ll)[0, 1]:.2f}")


# ---- experiment: what would a longest-lived-first dispatch of the backward's blocks buy?  (--lpt)
# The measured wave lives of the run above give an ORACLE order; the census build takes it as a launch-slot -> block table.
if "--lpt" in sys.argv:
    arr = np.zeros((nw["bwd"], 4), np.uint64)
    assert fn["bwd"](arr.ctypes.data_as(C.c_void_p), nw["bwd"], 0) == 0
    life = (arr[:, 2].astype(np.int64) - arr[:, 1].astype(np.int64)).astype(np.float64)
    nblk = nw["bwd"]

    def timed(order, label):
        od = torch.as_tensor(order.astype(np.int32)).cuda() if order is not None else None
        assert L.gsr_debug_bwd_order(C.c_void_p(od.data_ptr() if od is not None else 0)) == 0
        spans = []
        for _ in range(4):
            assert fn["bwd"](None, nw["bwd"], 1) == 0
            step()
            torch.cuda.synchronize()
            a = np.zeros((nw["bwd"], 4), np.uint64)
            assert fn["bwd"](a.ctypes.data_as(C.c_void_p), nw["bwd"], 0) == 0
            ok = a[:, 1] > 0
            spans.append((a[ok, 2].max() - a[ok, 1].min()) / 100.0)
        print(f"   {label:60s} span us: " + " ".join(f"{x:.1f}" for x in spans))
        assert L.gsr_debug_bwd_order(C.c_void_p(0)) == 0

    print("== backward blend, dispatch-order experiment (GSR_BWD_XCD as set in the environment applies only to the first line)")
    timed(None, "as shipped")
    ident = np.arange(nblk)
    timed(ident, "identity table (cost of the indirection)")
    timed(np.argsort(-life, kind="stable"), "longest-lived first, globally")
    # bands kept: launch slot s runs on XCD s % 8, position s // 8 in its queue; XCD x keeps its band of blocks, longest first
    per = nblk // 8
    tab = np.empty(nblk, np.int64)
    for x in range(8):
        band = np.arange(x * per, (x + 1) * per)
        tab[x::8] = band[np.argsort(-life[band], kind="stable")]
    timed(tab, "longest-lived first inside each XCD's band")
    tab2 = np.empty(nblk, np.int64)
    for x in range(8):
        tab2[x::8] = np.arange(x * per, (x + 1) * per)
    timed(tab2, "bands, natural order (= GSR_BWD_XCD=1 through the table)")
    # staggered starts: the first round of a band (1024 wave slots per XCD) ends all at once when it is uniformly heavy (a dip to 22
    # of 32 resident waves per CU at 40 % of the span): every second / fourth slot of the first round takes a LIGHT block instead
    for every in (2, 4):
        tabs = np.empty(nblk, np.int64)
        for x in range(8):
            band = np.arange(x * per, (x + 1) * per)
            o = list(band[np.argsort(-life[band], kind="stable")])
            first = min(1024, len(o))
            n_light = first // every
            heavy, light = o[:len(o) - n_light], o[len(o) - n_light:][::-1]
            seq, hi, li = [], 0, 0
            for k in range(first):
                if k % every == every - 1 and li < len(light):
                    seq.append(light[li]); li += 1
                else:
                    seq.append(heavy[hi]); hi += 1
            seq += heavy[hi:] + light[li:]
            tabs[x::8] = np.asarray(seq)
        timed(tabs, f"longest first inside each band, every {every}th slot of the first round a light block")
    # coarse classes only (what a cheap on-device binning could deliver): 8 classes by life
    cls = np.minimum(7, (8 * np.argsort(np.argsort(-life)) // nblk))
    tab3 = np.empty(nblk, np.int64)
    for x in range(8):
        band = np.arange(x * per, (x + 1) * per)
        tab3[x::8] = band[np.argsort(cls[band], kind="stable")]
    timed(tab3, "8 cost classes inside each band, natural order inside a class")

    # ---- predictors the forward could hand over: how well do they rank the blocks?
    masks = buf["point_list"]._gsr_block_masks[0].cpu().numpy()
    ncon = buf["n_contrib"].cpu().numpy()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    hits_wave = np.zeros(nblk, np.int64)     # mask hits over the 256-entry batches the forward WAVE (8x8 pixels) of the block walks
    hits_staged = np.zeros(nblk, np.int64)   # mask hits over what the forward staged (whole 256-entry batches up to the tile's last contributor)
    hits_kept = np.zeros(nblk, np.int64)     # mask hits up to the block's own last contributor (what the backward's compaction keeps)
    pix_sum = np.zeros(nblk, np.int64)       # sum over the block's pixels of n_contrib
    for tile in range(gx * gy):
        ty, tx = divmod(tile, gx)
        s0, s1 = ranges[tile]
        nc = ncon[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        if s1 <= s0 or nc.size == 0:
            continue
        ext = min(s1 - s0, 256 * (int(nc.max()) // 256 + 1))
        m = masks[s0:s0 + ext]
        for b in range(8):
            bx, by = b & 1, b >> 1
            blk = nc[by * 4:by * 4 + 4, bx * 8:bx * 8 + 8]
            bit = (m >> b) & 1
            hits_staged[tile * 8 + b] = int(bit.sum())
            hits_kept[tile * 8 + b] = int(bit[:int(blk.max()) if blk.size else 0].sum())
            pix_sum[tile * 8 + b] = int(blk.sum())
            wv = nc[(by >> 1) * 8:(by >> 1) * 8 + 8, bx * 8:bx * 8 + 8]           # the forward wave's 8x8 block
            hits_wave[tile * 8 + b] = int(bit[:min(s1 - s0, 256 * (int(wv.max()) // 256 + 1))].sum()) if wv.size else 0
    rank = lambda v: np.argsort(np.argsort(v))
    for nm, f in (("mask hits over the batches the forward wave walks", hits_wave), ("the same, in 16 geometric classes (2 per octave)", np.floor(2 * np.log2(np.maximum(1, hits_wave)))),
                  ("mask hits over the staged batches", hits_staged), ("mask hits up to the block's last contributor", hits_kept),
                  ("sum of n_contrib over the block's pixels", pix_sum), ("tile list length", np.repeat(list_len, 8))):
        rho = np.corrcoef(rank(f), rank(life))[0, 1]
        cls = np.minimum(7, (8 * rank(-f) // nblk))
        tab = np.empty(nblk, np.int64)
        for x in range(8):
            band = np.arange(x * per, (x + 1) * per)
            tab[x::8] = band[np.argsort(cls[band], kind="stable")]
        timed(tab, f"8 classes by [{nm}] (rank corr. with life {rho:.2f})")


# ---- experiment: what would a tile order for the FORWARD blend buy?  (--fwd-order)
# Oracle: the measured tile lives of the run above (max over the tile's four waves).  Orders that keep row-major order INSIDE each
# class (neighbouring tiles share their records in L2: a fully sorted order cost +23 us in round 3): the K lightest tiles last
# (K = tiles beyond the 2048 workgroup slots: they become the second round), the heaviest quarter first, both.
if "--fwd-order" in sys.argv:
    arr = np.zeros((nw["fwd"], 4), np.uint64)
    assert fn["fwd"](arr.ctypes.data_as(C.c_void_p), nw["fwd"], 0) == 0
    life = (arr[:, 2].astype(np.int64) - arr[:, 1].astype(np.int64)).astype(np.float64).reshape(tiles, 4).max(axis=1)
    slots = 2048

    def timed_fwd(order, label):
        od = torch.as_tensor(order.astype(np.int32)).cuda() if order is not None else None
        assert L.gsr_debug_fwd_order(C.c_void_p(od.data_ptr() if od is not None else 0)) == 0
        spans = []
        for _ in range(5):
            assert fn["fwd"](None, nw["fwd"], 1) == 0
            step()
            torch.cuda.synchronize()
            a = np.zeros((nw["fwd"], 4), np.uint64)
            assert fn["fwd"](a.ctypes.data_as(C.c_void_p), nw["fwd"], 0) == 0
            ok = a[:, 1] > 0
            spans.append((a[ok, 2].max() - a[ok, 1].min()) / 100.0)
        print(f"   {label:72s} span us: " + " ".join(f"{x:.1f}" for x in spans))

    ids = np.arange(tiles)
    K = max(0, tiles - slots)
    rank = np.argsort(np.argsort(life))              # 0 = lightest
    light = rank < K
    heavy = rank >= tiles - tiles // 4
    print(f"== forward tile order experiment ({name}): {tiles} tiles, {slots} slots, K = {K}; tile life us p10 {np.percentile(life, 10) / 100:.1f} p50 {np.percentile(life, 50) / 100:.1f} p90 {np.percentile(life, 90) / 100:.1f}")
    timed_fwd(None, "row-major (the product)")
    timed_fwd(ids, "row-major through the table (the table's own cost)")
    timed_fwd(np.concatenate([ids[~light], ids[light]]), "the K lightest tiles last, row-major inside both classes")
    timed_fwd(np.concatenate([ids[heavy], ids[~heavy & ~light], ids[light]]), "heaviest quarter first, lightest K last, row-major inside the classes")
    timed_fwd(np.argsort(-life, kind="stable"), "fully sorted, heaviest first (no locality)")
    # a predictor that needs no previous frame: the tile's list length
    lrank = np.argsort(np.argsort(list_len))
    timed_fwd(np.concatenate([ids[lrank >= K], ids[lrank < K]]), "the K tiles with the SHORTEST LISTS last (no oracle)")
    # is it the prediction at all, or just that ANY order which is not row-major spreads neighbouring (similar) tiles over the CUs?
    rng = np.random.default_rng(3)
    timed_fwd(rng.permutation(tiles), "a random permutation (no cost information at all)")
    for stride in (389, 577, 1009):
        timed_fwd((ids * stride) % tiles if np.gcd(stride, tiles) == 1 else ids, f"stride permutation, tile = slot * {stride} mod tiles")
    gx = (W + 15) // 16
    gy = tiles // gx
    timed_fwd(np.concatenate([np.arange(r, gy, 8) for r in range(8)])[:, None].repeat(gx, 1).__mul__(gx).__add__(np.arange(gx)[None, :]).reshape(-1), "tile rows interleaved by 8 (row r, r + 8, ...)")
    # what the entry-count predictors would see (4 ints per tile: walked << 16 | staged): needs a census build made with
    # EXTRA=-DGSR_FWD_COST_LIFE=0 -- the product's waves leave their measured life in the `walked` field since late round 4
    host = importlib.import_module("3dgs-native_amd._host")
    ws = [t for (kind, _, _), t in host._ws.items() if kind == "bin"][0]
    raw = ws[:16 * tiles].view(torch.int32).cpu().numpy().reshape(tiles, 4)
    walked, staged = (raw >> 16) & 0x7FFF, raw & 0xFFFF
    cands = {"max walked": walked.max(1), "max staged": staged.max(1), "max (walked + staged / 8)": (walked + staged // 8).max(1),
             "sum walked": walked.sum(1), "max walked + staged / 2": walked.max(1) + staged.max(1) // 2, "max walked + 2 staged": walked.max(1) + 2 * staged.max(1)}
    for label, c in cands.items():
        r = np.corrcoef(np.argsort(np.argsort(c)), np.argsort(np.argsort(life)))[0, 1]
        cls = 63 - (c.astype(np.int64) * 63 // max(1, int(c.max())))
        timed_fwd(np.argsort(cls, kind="stable"), f"64 classes of {label} (rank correlation with the life {r:.2f})")
    assert L.gsr_debug_fwd_order(C.c_void_p(0)) == 0
