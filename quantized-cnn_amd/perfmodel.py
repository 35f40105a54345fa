"""Work counts of the two hot kernels, restated from quantized-cnn_amd/csrc (qk_conv_slots / qk_conv_aprx /
k_conv_aprx / k_fc_aprx) so that bench.py can turn a measured layer time into the figures DESIGN.md §3 argues
with: LUT stages built, look-ups per stage, how often every source pixel's table is rebuilt, matrix-pipe and LDS
read utilisation.  Pure arithmetic — nothing here touches a device.
"""
from __future__ import annotations

from .topology import CONV, FCNT

PANEL = 128
GATHER_WAVES = 12
CUS = 256
CLOCK_HZ = 2.4e9                       # MI355X_MICROARCH.md: max shader clock
F32_MFMA_FLOPS = 157.3e12              # dense f32 matrix peak (v_mfma_f32_16x16x4_f32: 256 FLOP/clk/CU)
LDS_READ_BYTES_PER_CLK = 256           # per CU (ds_read_b64 / ds_read_b128)


def conv_tile(ctg: int):
    """(TH, TW, channels per gather wave, workgroups along the channel axis) for `ctg` output channels per group."""
    chunks = (ctg + 383) // 384
    per = (ctg + chunks - 1) // chunks
    cpw = 4 if per <= 48 else 6 if per <= 72 else 8 if per <= 96 else 12 if per <= 144 else 16 if per <= 192 else 24 if per <= 288 else 32
    th, tw = {32: (1, 1), 24: (1, 1), 16: (1, 2), 12: (1, 3), 8: (2, 2), 6: (2, 3), 4: (2, 4)}[cpw]
    chunks = (ctg + GATHER_WAVES * cpw - 1) // (GATHER_WAVES * cpw)
    return th, tw, cpw, chunks


def stage_group(k: int) -> int:
    return 128 // k if k <= 64 else 1


def sym8_tile(ctg: int):
    """(TH, TW, channels per wave, channel chunks) of the eight-wave symmetric kernel (qk_conv_sym8_config)."""
    chunks = (ctg + 383) // 384
    per = (ctg + chunks - 1) // chunks
    cpw, th, tw = (16, 2, 3) if per <= 128 else (24, 2, 2) if per <= 192 else (32, 1, 3) if per <= 256 else (48, 1, 2)
    return th, tw, cpw, (ctg + 8 * cpw - 1) // (8 * cpw)


def half8_tile(ctg: int):
    """(TH, TW, channels per wave, wave sets, channel chunks) of the half-panel eight-wave kernel (qk_conv_half8_config)."""
    chunks = (ctg + 511) // 512
    cpw, th, tw, ws = {128: (32, 3, 4, 2), 192: (48, 2, 4, 2), 256: (32, 2, 3, 1), 384: (48, 2, 2, 1), 512: (64, 1, 3, 1)}[ctg // chunks]
    return th, tw, cpw, ws, chunks


def half8_slide_cfg(ctg: int):
    """(slots, columns per strip, channels per wave, wave sets, channel chunks) of its sliding form (qk_conv_half8_slide_config)."""
    _, _, cpw, ws, chunks = half8_tile(ctg)
    return 3, {128: 4, 192: 2, 256: 2, 384: 1, 512: 1}[ctg // chunks], cpw, ws, chunks


def conv_work(in_hwc, out_hwc, ly, m: int, k: int, cs: int, sym=False):
    """Per 128-image panel: stages built, look-ups (border clipped), ideal stages (every (pixel, sub-space group)
    once per group), f32 MFMA FLOP issued.  sym: True = the 16-wave symmetric kernel's 2x2 tile of all 128 channels
    (k_conv_sym), 8 = the eight-wave symmetric kernel's tile (k_conv_sym8), "h8" = the half-panel eight-wave kernel
    (k_conv_half8): its workgroups hold 64 images, so `stages` counts TWO half-panel stages per tile stage and a stage issues half
    the matrix work ("half_stages": True)."""
    h, w, cin = in_hwc
    ho, wo, ct = out_hwc
    knl, s, p, grp = ly["knl"], ly["stride"], ly["pad"], ly["grp"]
    ws = 1
    if sym == "h8":
        th, tw, cpw, ws, chunks = half8_tile(ct // grp)
    else:
        th, tw, cpw, chunks = sym8_tile(ct // grp) if sym == 8 else (2, 2, 8, 1) if sym else conv_tile(ct // grp)
    g = stage_group(k)
    mg = (m + g - 1) // g
    stages = 0
    for ty in range((ho + th - 1) // th):
        ho0, hol = ty * th, min(ty * th + th, ho) - 1
        rows = min(h - 1, hol * s - p + knl - 1) - max(0, ho0 * s - p) + 1
        for tx in range((wo + tw - 1) // tw):
            wo0, wol = tx * tw, min(tx * tw + tw, wo) - 1
            cols = min(w - 1, wol * s - p + knl - 1) - max(0, wo0 * s - p) + 1
            stages += rows * cols * mg
    stages *= chunks * grp
    taps_h = sum(min(knl - 1, h - 1 - (o * s - p)) - max(0, -(o * s - p)) + 1 for o in range(ho))
    taps_w = sum(min(knl - 1, w - 1 - (o * s - p)) - max(0, -(o * s - p)) + 1 for o in range(wo))
    lookups = taps_h * taps_w * m * ct                      # per image
    ideal = h * w * mg * grp
    ks = 2 if min(cin // grp, cs) > 4 else 1
    # algorithmic LUT build (SURVEY.md §8 table "LUT-MAC/img"): every (pixel, sub-space) table once, over the dims it has
    alg_flop = 2 * h * w * grp * k * sum(min(cs, cin // grp - i * cs) for i in range(m)) * 128
    if sym == "h8":
        return dict(stages=2 * stages, lookups=lookups, ideal_stages=ideal, mfma_flop=2 * stages * 128 * 64 * 4 * ks * 2, alg_flop=alg_flop,
                    half_stages=True, tile="half panels, 8 waves %dx%dx%d (%d wave set%s)" % (th, tw, 8 // ws * cpw, ws, "s" if ws > 1 else ""), ks=ks)
    return dict(stages=stages, lookups=lookups, ideal_stages=ideal, mfma_flop=stages * 128 * 128 * 4 * ks * 2, alg_flop=alg_flop,
                tile=("symmetric 8 waves %dx%dx%d" % (th, tw, 8 * cpw)) if sym == 8 else
                     ("symmetric %dx%dx%d" % (th, tw, 16 * cpw)) if sym else "%dx%dx%d" % (th, tw, GATHER_WAVES * cpw), ks=ks)


def conv_work_slide(in_hwc, out_hwc, ly, m: int, k: int, cs: int, seg_beg):
    """The same counts for the sliding kernel (k_conv_aprx<.., SLIDE>): every output column is cut into the row segments
    [seg_beg[i], seg_beg[i + 1]); a workgroup builds the source rows under its segment x the knl columns of its window."""
    h, w, cin = in_hwc
    ho, wo, ct = out_hwc
    knl, s, p, grp = ly["knl"], ly["stride"], ly["pad"], ly["grp"]
    base = conv_work(in_hwc, out_hwc, ly, m, k, cs)
    _, _, cpw, chunks = conv_tile(ct // grp)
    g = stage_group(k)
    mg = (m + g - 1) // g
    slots = (knl + s - 1) // s
    nc = 2 if (slots == 3 and cpw <= 6) else 1          # output columns per strip (qk_slide_config)
    stages = 0
    for x0 in range(0, wo, nc):
        x1 = min(wo, x0 + nc) - 1
        cols = min(w - 1, x1 * s - p + knl - 1) - max(0, x0 * s - p) + 1
        for a, b in zip(seg_beg[:-1], seg_beg[1:]):
            rows = min(h - 1, (b - 1) * s - p + knl - 1) - max(0, a * s - p) + 1
            stages += max(rows, 0) * max(cols, 0) * mg
    stages *= chunks * grp
    return dict(base, stages=stages, mfma_flop=stages * 128 * 128 * 4 * base["ks"] * 2,
                tile="slide %d column(s) x %d slots x %d, %d segment(s) per column" % (nc, slots, GATHER_WAVES * cpw, len(seg_beg) - 1))


def sym8_slide_cfg(ctg: int, knl: int, stride: int):
    """(slots, columns per strip, channels per wave, channel chunks) of the sliding eight-wave kernel (qk_conv_sym8_slide_config)."""
    ns = (knl + stride - 1) // stride
    chunks = (ctg + 255) // 256
    per = (ctg + chunks - 1) // chunks
    if ns == 3:
        cpw, nc = (16, 2) if per <= 128 else (24, 1) if per <= 192 else (32, 1)
    else:
        cpw, nc = 16, 1
    return ns, nc, cpw, (ctg + 8 * cpw - 1) // (8 * cpw)


def conv_work_slide8(in_hwc, out_hwc, ly, m: int, k: int, cs: int, seg_beg):
    """conv_work_slide for k_conv_sym8<.., SLIDE>: strips of nc output columns, row segments [seg_beg[i], seg_beg[i + 1])."""
    h, w, cin = in_hwc
    ho, wo, ct = out_hwc
    knl, s, p, grp = ly["knl"], ly["stride"], ly["pad"], ly["grp"]
    base = conv_work(in_hwc, out_hwc, ly, m, k, cs)
    ns, nc, cpw, chunks = sym8_slide_cfg(ct // grp, knl, s)
    stages = 0
    for x0 in range(0, wo, nc):
        x1 = min(wo, x0 + nc) - 1
        cols = min(w - 1, x1 * s - p + knl - 1) - max(0, x0 * s - p) + 1
        for a, b in zip(seg_beg[:-1], seg_beg[1:]):
            rows = min(h - 1, (b - 1) * s - p + knl - 1) - max(0, a * s - p) + 1
            stages += max(rows, 0) * max(cols, 0) * m
    stages *= chunks * grp
    return dict(base, stages=stages, mfma_flop=stages * 128 * 128 * 4 * base["ks"] * 2,
                tile="slide 8 waves %d column(s) x %d slots x %d, %d segment(s) per column" % (nc, ns, 8 * cpw, len(seg_beg) - 1))


def conv_work_slide_h8(in_hwc, out_hwc, ly, m: int, k: int, cs: int, seg_beg):
    """conv_work_slide for k_conv_half8<.., SLIDE>: strips of nc output columns, row segments, TWO half-panel stages per strip stage."""
    h, w, cin = in_hwc
    ho, wo, ct = out_hwc
    knl, s, p, grp = ly["knl"], ly["stride"], ly["pad"], ly["grp"]
    base = conv_work(in_hwc, out_hwc, ly, m, k, cs, "h8")
    ns, nc, cpw, ws, chunks = half8_slide_cfg(ct // grp)
    stages = 0
    for x0 in range(0, wo, nc):
        x1 = min(wo, x0 + nc) - 1
        cols = min(w - 1, x1 * s - p + knl - 1) - max(0, x0 * s - p) + 1
        for a, b in zip(seg_beg[:-1], seg_beg[1:]):
            rows = min(h - 1, (b - 1) * s - p + knl - 1) - max(0, a * s - p) + 1
            stages += max(rows, 0) * max(cols, 0) * m
    stages *= chunks * grp * 2
    return dict(base, stages=stages, mfma_flop=stages * 128 * 64 * 4 * base["ks"] * 2,
                tile="half panels, slide 8 waves %d column(s) x %d slots x %d, %d segment(s) per column" % (nc, ns, 8 // ws * cpw, len(seg_beg) - 1))


def fc_work(d: int, ct: int, m: int, k: int, cs: int, msplit_chunks: int):
    g = stage_group(k)
    stages = (m + g - 1) // g * msplit_chunks
    ks = 2 if min(d, cs) > 4 else 1
    alg_flop = 2 * k * sum(min(cs, d - i * cs) for i in range(m)) * 128
    return dict(stages=stages, lookups=m * ct, ideal_stages=(m + g - 1) // g, mfma_flop=stages * 128 * 128 * 4 * ks * 2,
                alg_flop=alg_flop, tile="fc", ks=ks)


def decoded_report(sizes, layers, l: int, images: float, ms: float, nchw: bool = False):
    """A conv layer that ran through its decoded code words (qcnn_decoded.hip): products issued on the matrix pipe against
    the dense f32 peak and against what `scripts/ubench/mfma_clock.hip` sustains on all CUs (145 TFLOP/s at the 2.29 GHz the
    chip holds under that load).  Panel kernel (k_conv_dec): kernel rows of knl * Cin products padded to fours, positions x
    channels x images.  nchw (k_conv_dec_nchw, the network input read in place): Cin * knl^2 products flat, padded to 16;
    output rows in groups of four positions, images in sixteens."""
    ly = layers[l]
    h, w, c = sizes[l]
    ho, wo, ct = sizes[l + 1]
    if nchw:
        kflat = (ly["knl"] * ly["knl"] * c + 15) // 16 * 16
        flop = 2.0 * ho * ((wo + 3) // 4 * 4) * ct * kflat * ((images + 15) // 16 * 16)
        tile = ("decoded code words, NCHW input in place: %d (of %d) products per output, 16 images x 4 positions x 96 channels per wave"
                % (kflat, ly["knl"] * ly["knl"] * c))
    else:
        kp = (ly["knl"] * c + 3) // 4 * 4
        flop = 2.0 * ho * wo * ct * ly["knl"] * kp * images
        tile = "decoded code words: %d x %d products per output, 64 images x %d channels per wave" % (
            ly["knl"], kp, 96 if ct % 96 == 0 else (64 if ct % 64 == 0 else 32))
    t = ms * 1e-3
    return dict(tile=tile,
                issued_mfma_flop_per_image=int(flop / images), mfma_util=round(flop / t / F32_MFMA_FLOPS, 4) if t > 0 else 0.0,
                mfma_util_of_sustained=round(flop / t / 145.0e12, 4) if t > 0 else 0.0,
                mfma_algorithmic_frac=round(2.0 * ho * wo * ct * ly["knl"] * ly["knl"] * c * images / t / F32_MFMA_FLOPS, 4) if t > 0 else 0.0,
                lookups_replaced_per_image=int(conv_work(sizes[l], sizes[l + 1], ly, 1, 128, c)["lookups"]))


def layer_report(sizes, layers, params, l: int, images: float, ms: float, seg_beg=None, sym=False):
    """Roofline-style figures of conv/FC layer l for a launch over `images` images that took `ms` milliseconds; seg_beg:
    the sliding kernel's row segments when the layer ran it (QcnnEngine.layer_segments)."""
    ly = layers[l]
    mm, kk, cc = (int(x) for x in params[l]["ctrd"].shape)
    if ly["type"] == CONV and seg_beg and sym == "h8":
        wk = conv_work_slide_h8(sizes[l], sizes[l + 1], ly, mm, kk, cc, seg_beg)
    elif ly["type"] == CONV and seg_beg and sym == 8:
        wk = conv_work_slide8(sizes[l], sizes[l + 1], ly, mm, kk, cc, seg_beg)
    elif ly["type"] == CONV and seg_beg:
        wk = conv_work_slide(sizes[l], sizes[l + 1], ly, mm, kk, cc, seg_beg)
    elif ly["type"] == CONV:
        wk = conv_work(sizes[l], sizes[l + 1], ly, mm, kk, cc, sym)
    elif ly["type"] == FCNT:
        e = sizes[l][0] * sizes[l][1] * sizes[l][2]
        cpw = 32 if ly["nod"] >= 384 else (8 if ly["nod"] >= 96 else 4)
        chunks = (ly["nod"] + GATHER_WAVES * cpw - 1) // (GATHER_WAVES * cpw)
        if sym == 8:                                        # k_fc_sym8: eight waves x 96 channels per workgroup
            chunks = (ly["nod"] + 767) // 768
        wk = fc_work(e, ly["nod"], mm, kk, cc, chunks)
        if sym == 8:
            wk["tile"] = "fc, 8 waves x 96 channels"
    else:
        return None
    panels = (images + PANEL - 1) // PANEL
    t = ms * 1e-3
    stages = wk["stages"] * panels
    cycles = t * CLOCK_HZ * CUS / stages if stages else 0.0
    lookups = wk["lookups"] * images
    half = 2 if wk.get("half_stages") else 1                # half-panel stages: two per panel, each with rows of 64 images
    return dict(tile=wk["tile"], stages_per_panel=wk["stages"], rebuild_factor=round(wk["stages"] / half / wk["ideal_stages"], 2),
                lookups_per_stage=round(wk["lookups"] * half / wk["stages"], 1),   # row look-ups (128 images each; half-panel kernel: 64) per built stage
                stage_cycles=round(cycles, 0),
                mfma_util=round(wk["mfma_flop"] * panels / t / F32_MFMA_FLOPS, 4) if t > 0 else 0.0,
                # the LUT build the algorithm asks for (every table once), on the images the launch really holds
                mfma_algorithmic_frac=round(wk["alg_flop"] * images / PANEL / t / F32_MFMA_FLOPS, 4) if t > 0 else 0.0,
                lookup_gbs=round(lookups * 4 / t / 1e9, 1) if t > 0 else 0.0,
                lds_frac=round(lookups * 4 / t / (LDS_READ_BYTES_PER_CLK * CUS * CLOCK_HZ), 4) if t > 0 else 0.0)
