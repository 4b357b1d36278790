"""numpy model of the index arithmetic of csrc/conv_igemm.hip + csrc/layout.hip (TEST INFRASTRUCTURE).

It restates, in vectorised numpy, exactly the address/tap/phase/de-slice formulas the HIP kernels use, so
that the *algorithm* (sub-pixel phase decomposition of the transposed conv, in-place skip concat, class
plane as border-aware bias, slice/de-slice maps, packed-weight order) can be checked against the oracle on
a CPU-only machine.  It says nothing about the MFMA lane maps or LDS staging -- those are covered by the
-m gpu parity tests.
"""
import numpy as np


def sep_slice_input(mix, masks=None):
    B, F, T, C = mix.shape
    Hs = F // 16
    x = mix if masks is None else np.log1p(np.maximum(masks * (np.exp(mix) - 1.0), 0.0))
    out = np.empty((B, Hs, T, 16 * C), np.float32)
    for n in range(16 * C):
        c, s = n >> 4, n & 15
        out[:, :, :, n] = x[:, s * Hs:(s + 1) * Hs, :, c]
    return out


def pack_conv_weight(w, ci_used=None):
    Co, Ci, KH, KW = w.shape
    cu = Ci if ci_used is None else ci_used
    return np.ascontiguousarray(w[:, :cu].transpose(0, 2, 3, 1)).reshape(Co, KH * KW * cu)


def pack_convT_weight(w):
    Ci, Co = w.shape[:2]
    wp = np.empty((4, Co, 2, 2, Ci), np.float32)
    for phase in range(4):
        ph, pw = phase >> 1, phase & 1
        for th in range(2):
            for tw in range(2):
                kh = (2 if ph else 1) + th * (-2 if ph else 2)
                kw = (2 if pw else 1) + tw * (-2 if pw else 2)
                wp[phase, :, th, tw, :] = w[:, :, kh, kw].T
    return wp.reshape(4, Co, 4 * Ci)


def class_table(w, plane):
    Co = w.shape[0]
    t = np.zeros((9, Co), np.float32)
    for ch in range(3):
        for cw in range(3):
            h0, h1 = (1 if ch == 0 else 0), (3 if ch == 2 else 4)
            w0, w1 = (1 if cw == 0 else 0), (3 if cw == 2 else 4)
            t[ch * 3 + cw] = w[:, plane, h0:h1, w0:w1].sum(axis=(1, 2))
    return t


def fold_bn(g, b, mean, var, eps):
    s = g / np.sqrt(var + eps)
    return s.astype(np.float32), (b - mean * s).astype(np.float32)


def conv_igemm(src0, src1, wp, N, Hq, Wq, stride, nth, ntw, mulh, offh, mulw, offw, conv_transpose,
               scale, shift, slope, cls_table, cls_val, Ho, Wo, os_, ph0, pw0, out_mode):
    """Mirror of igemm_f32_kernel.  src*: NHWC arrays.  Returns dst (NHWC [B,Ho,Wo,N] or BHWC de-sliced)."""
    B, Hi, Wi, C0 = src0.shape
    C1 = 0 if src1 is None else src1.shape[3]
    Ctot = C0 + C1
    K = nth * ntw * Ctot
    if out_mode == 0:
        dst = np.zeros((B, Ho, Wo, N), np.float32)
    else:
        dst = np.zeros((B, 16 * Ho, Wo, N // 16), np.float32)
    m = np.arange(B * Hq * Wq)
    r = m % Wq
    q = (m // Wq) % Hq
    b = m // (Wq * Hq)
    for phase in range(4 if conv_transpose else 1):
        if conv_transpose:
            ph, pw = phase >> 1, phase & 1
            mh, mw, oh_, ow_ = 2 * ph - 1, 2 * pw - 1, 0, 0
            w = wp[phase]
        else:
            ph, pw, mh, mw, oh_, ow_ = ph0, pw0, mulh, mulw, offh, offw
            w = wp
        A = np.zeros((m.size, K), np.float32)
        for tap in range(nth * ntw):
            th, tw = tap // ntw, tap % ntw
            ih = q * stride + oh_ + th * mh
            iw = r * stride + ow_ + tw * mw
            ok = (ih >= 0) & (ih < Hi) & (iw >= 0) & (iw < Wi)
            ihc, iwc = np.clip(ih, 0, Hi - 1), np.clip(iw, 0, Wi - 1)
            v0 = src0[b, ihc, iwc, :] * ok[:, None]
            A[:, tap * Ctot: tap * Ctot + C0] = v0
            if C1:
                A[:, tap * Ctot + C0:(tap + 1) * Ctot] = src1[b, ihc, iwc, :] * ok[:, None]
        D = A @ w.reshape(N, K).T
        oh = q * os_ + ph
        ow = r * os_ + pw
        if cls_table is not None:
            ch = np.where(oh == 0, 0, np.where(oh == Ho - 1, 2, 1))
            cw = np.where(ow == 0, 0, np.where(ow == Wo - 1, 2, 1))
            D = D + cls_val[b][:, None] * cls_table[ch * 3 + cw]
        if scale is not None:
            D = D * scale[None, :]
        if shift is not None:
            D = D + shift[None, :]
        D = np.where(D > 0, D, D * slope).astype(np.float32)
        if out_mode == 0:
            dst[b, oh, ow, :] = D
        else:
            for n in range(N):
                c, s = n >> 4, n & 15
                dst[b, s * Ho + oh, ow, c] = D[:, n]
    return dst


def unet_forward(sd, enc_pre, dec_pre, mix, target_class=None, masks=None, eps=1e-5):
    """Whole U-Net through the kernel model; sd: numpy state dict."""
    x = sep_slice_input(mix, masks)
    feats = []
    for i in range(5):
        w = sd[enc_pre + "%d.0.weight" % i]
        first_bin = (i == 0 and masks is None)
        wp = pack_conv_weight(w, 32 if i == 0 else None)
        sc, sh = fold_bn(sd[enc_pre + "%d.1.weight" % i], sd[enc_pre + "%d.1.bias" % i],
                         sd[enc_pre + "%d.1.running_mean" % i], sd[enc_pre + "%d.1.running_var" % i], eps)
        tab = class_table(w, 32) if first_bin else None
        cv = (target_class.reshape(-1).astype(np.float32) + 1.0) if first_bin else None
        B, H, W, _ = x.shape
        x = conv_igemm(x, None, wp, w.shape[0], H // 2, W // 2, 2, 4, 4, 1, -1, 1, -1, False, sc, sh, 0.2, tab, cv,
                       H // 2, W // 2, 1, 0, 0, 0)
        feats.append(x)
    out = feats[4]
    skips = feats[:4][::-1]
    for i in range(5):
        w = sd[dec_pre + "%d.0.weight" % i]
        sc, sh = fold_bn(sd[dec_pre + "%d.1.weight" % i], sd[dec_pre + "%d.1.bias" % i],
                         sd[dec_pre + "%d.1.running_mean" % i], sd[dec_pre + "%d.1.running_var" % i], eps)
        B, H, W, _ = out.shape
        out = conv_igemm(out, None if i == 0 else skips[i - 1], pack_convT_weight(w), w.shape[1], H, W, 1, 2, 2,
                         0, 0, 0, 0, True, sc, sh, 0.0, None, None, 2 * H, 2 * W, 2, 0, 0, 0)
    hw = sd[dec_pre + "5.0.weight"]
    B, H, W, _ = out.shape
    return conv_igemm(out, None, pack_conv_weight(hw), hw.shape[0], H, W, 1, 1, 1, 0, 0, 0, 0, False, None,
                      sd[dec_pre + "5.0.bias"], 1.0, None, None, H, W, 1, 0, 0, 1), feats


# ---------------------------------------------------------------------------------------------------------------------
# bf16x3 math and the split32 layout (csrc/conv_igemm.hip SPLIT modes, include/m2h.h)
# ---------------------------------------------------------------------------------------------------------------------
def bf16_rne(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, as v_cvt_pk_bf16_f32 / a (__bf16) cast."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split_hi_lo(x):
    hi = bf16_rne(x)
    lo = bf16_rne(np.asarray(x, np.float32) - hi)
    return hi, lo


def split32(x):
    """fp32 array (last dim % 32 == 0) -> uint16 array [..., groups, 2, 32]: per 32-value group, 32 hi halves then 32 lo halves
    (the byte image m2h_split32 writes in place of the group's 128 bytes)."""
    x = np.asarray(x, np.float32)
    g = x.reshape(x.shape[:-1] + (x.shape[-1] // 32, 32))
    hi, lo = split_hi_lo(g)
    return np.stack([(hi.view(np.uint32) >> 16).astype(np.uint16), (lo.view(np.uint32) >> 16).astype(np.uint16)], axis=-2)


def bf16x3_matmul(a, w):
    """A [M,K] x W[N,K]^T with the kernel's product formation: a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulation
    (numpy accumulates in float64 here: the model bounds the PRODUCT error, not the summation order)."""
    ah, al = split_hi_lo(a)
    wh, wl = split_hi_lo(w)
    f = np.float64
    return ah.astype(f) @ wh.astype(f).T + ah.astype(f) @ wl.astype(f).T + al.astype(f) @ wh.astype(f).T


def convT_tap_phase(x, wp_phase, ph, pw, bm=128):
    """Model of convT_tap_kernel for ONE sub-pixel phase: x NHWC [B,H,W,C] (one source), wp_phase [N][4*C] packed as
    pack_convT_weight()[phase].  Follows the kernel's data flow: per tile of `bm` output pixels the (R+1) x (W+1) input
    pixels are staged once (rows q0+hoff .., cols woff ..), and tap (th,tw) reads the window shifted by
    tapoff = (th*dh - hoff)*W1 + (tw*dw - woff).  Returns the phase's outputs [B,H,W,N] (pixel (q,r) -> output (2q+ph, 2r+pw))."""
    B, H, W, C = x.shape
    N = wp_phase.shape[0]
    assert bm % W == 0 and H % (bm // W) == 0
    R, W1 = bm // W, W + 1
    dh, dw = 2 * ph - 1, 2 * pw - 1
    hoff, woff = min(dh, 0), min(dw, 0)
    out = np.zeros((B, H, W, N), np.float64)
    wt = wp_phase.reshape(N, 2, 2, C).astype(np.float64)
    for b in range(B):
        for q0 in range(0, H, R):
            img = np.zeros(((R + 1) * W1, C), np.float64)          # the staged LDS image
            for l in range((R + 1) * W1):
                qi, rr = divmod(l, W1)
                ih, iw = q0 + qi + hoff, rr + woff
                if 0 <= ih < H and 0 <= iw < W:
                    img[l] = x[b, ih, iw]
            for ml in range(bm):                                    # tile row -> (qi, r); LDS row = qi*W1 + r + tapoff
                qi, r = divmod(ml, W)
                acc = np.zeros(N)
                for th in range(2):
                    for tw in range(2):
                        tapoff = (th * dh - hoff) * W1 + (tw * dw - woff)
                        acc += wt[:, th, tw, :] @ img[qi * W1 + r + tapoff]
                out[b, q0 + qi, r] = acc
    return out


def tap_range(ntaps, Q, stride, off, mul, extent):
    """Mirror of tap_range() in csrc/conv_igemm.hip: first tap and count of the contiguous run of kernel rows (columns) that reach
    inside the image for at least one of the Q output rows (columns)."""
    valid = [t for t in range(ntaps) if any(0 <= q * stride + off + t * mul < extent for q in range(Q))]
    if not valid:
        return 0, ntaps
    return valid[0], valid[-1] - valid[0] + 1


def conv_with_tap_window(x, wp, N, stride, nth, ntw, off, Hq, Wq):
    """Forward conv (mul 1) walking only the tap window, as the scalar-decode loader does: the skipped taps must be exact zeros.
    x NHWC [B,Hi,Wi,C], wp [N][nth*ntw*C]."""
    B, Hi, Wi, C = x.shape
    th0, thn = tap_range(nth, Hq, stride, off, 1, Hi)
    tw0, twn = tap_range(ntw, Wq, stride, off, 1, Wi)
    out = np.zeros((B, Hq, Wq, N), np.float64)
    for th in range(th0, th0 + thn):
        for tw in range(tw0, tw0 + twn):
            wk = wp[:, (th * ntw + tw) * C:(th * ntw + tw + 1) * C].astype(np.float64)
            for q in range(Hq):
                ih = q * stride + off + th
                if not 0 <= ih < Hi:
                    continue
                for r in range(Wq):
                    iw = r * stride + off + tw
                    if 0 <= iw < Wi:
                        out[:, q, r, :] += x[:, ih, iw, :].astype(np.float64) @ wk.T
    return out, (th0, thn, tw0, twn)


def conv3x3_row(x, wp, N, mul):
    """Mirror of conv3x3_row_kernel's indexing (csrc/conv_igemm.hip): chunks of four image rows staged as a zero-padded 6 x 34
    patch; tap t = (th, tw) reads the patch at the output pixel's own position shifted by (off + th*mul, off + tw*mul),
    off = -mul (forward: mul 1; input gradient: mul -1).  x NHWC [B,H,32,C], wp [N][9*C]."""
    B, H, W, C = x.shape
    assert W == 32 and H % 4 == 0
    PW = W + 2
    off = -mul
    out = np.zeros((B, H, W, N), np.float64)
    for b in range(B):
        for q0 in range(0, H, 4):
            patch = np.zeros((6 * PW, C), np.float64)
            for pr in range(6):
                ih = q0 + pr - 1
                if 0 <= ih < H:
                    patch[pr * PW + 1:pr * PW + 1 + W] = x[b, ih]
            for wave in range(4):
                prow0 = (wave + 1) * PW + 1
                for t in range(9):
                    shift = (off + (t // 3) * mul) * PW + (off + (t % 3) * mul)
                    a = patch[prow0 + shift:prow0 + shift + W]                      # [32 pixels][C]
                    out[b, q0 + wave] += a @ wp[:, t * C:(t + 1) * C].astype(np.float64).T
    return out


def wgrad3x3_row(x, dy):
    """Mirror of wgrad3x3_row_kernel's indexing (csrc/conv_bwd.hip): one image row per chunk staged as a zero-padded 3 x 34 patch,
    each of the four waves takes 8 pixels, tap (ty, tx) reads patch row ty at column pixel + tx; -> dW packed [N][9*C]."""
    B, H, W, C = x.shape
    N = dy.shape[3]
    assert W == 32
    PW = W + 2
    dw = np.zeros((N, 9 * C), np.float64)
    for b in range(B):
        for q in range(H):
            patch = np.zeros((3 * PW, C), np.float64)
            for pr in range(3):
                ih = q + pr - 1
                if 0 <= ih < H:
                    patch[pr * PW + 1:pr * PW + 1 + W] = x[b, ih]
            for wave in range(4):
                m = np.arange(wave * 8, wave * 8 + 8)
                for t in range(9):
                    ty, tx = t // 3, t % 3
                    dw[:, t * C:(t + 1) * C] += dy[b, q, m].astype(np.float64).T @ patch[ty * PW + m + tx]
    return dw


def wgrad3x3_row_bf16x3(x, dy, splits):
    """Mirror of wgrad3x3_row_bf16x3_kernel's indexing (csrc/conv_bwd.hip, round 4), in bf16x3 arithmetic: a split walks its image rows
    c0 .. c1-1 of the flattened (b, row) list; x rows live TRANSPOSED ([channel][32 pixels]) in a ring of four slots (slot = row & 3: step c
    reads slots of rows c-1, c, c+1 while row c+2 arrives), rows outside the image are skipped; the tap's column shift is applied to
    dY -- dW[n][ty][tx][c] += sum_px' dY[px' - tx + 1][n] * x[row + ty - 1][px'][c], zeros shifted in at the row ends -- and one product is
    lo*hi + hi*lo + hi*hi of the bf16 splits.  Returns the per-split slabs [splits][N][9*C]."""
    B, H, W, C = x.shape
    N = dy.shape[3]
    assert W == 32
    rows_total = B * H
    xr, yr = x.reshape(rows_total, W, C), dy.reshape(rows_total, W, N)
    slabs = np.zeros((splits, N, 9 * C), np.float64)
    for s in range(splits):
        c0, c1 = (rows_total * s) // splits, (rows_total * (s + 1)) // splits
        ring = [None] * 4
        for r in (c0 - 1, c0, c0 + 1):                       # prologue
            if 0 <= r < rows_total:
                ring[r & 3] = (r, split_hi_lo(xr[r].T))       # [C][32] hi / lo
        for c in range(c0, c1):
            if 0 <= c + 2 < rows_total:
                ring[(c + 2) & 3] = (c + 2, split_hi_lo(xr[c + 2].T))
            q = c % H
            yh, yl = split_hi_lo(yr[c].T)                     # [N][32]
            for ty in range(3):
                if not 0 <= q + ty - 1 < H:
                    continue
                tag, (xh, xl) = ring[(c + ty - 1) & 3]
                assert tag == c + ty - 1, "ring slot overwritten before use"
                for tx in range(3):
                    sh = 1 - tx                               # element k takes dY[k + sh]
                    ah, al = np.zeros_like(yh), np.zeros_like(yl)
                    if sh == 1:
                        ah[:, :-1], al[:, :-1] = yh[:, 1:], yl[:, 1:]
                    elif sh == -1:
                        ah[:, 1:], al[:, 1:] = yh[:, :-1], yl[:, :-1]
                    else:
                        ah, al = yh, yl
                    t = ty * 3 + tx
                    slabs[s, :, t * C:(t + 1) * C] += (al.astype(np.float64) @ xh.astype(np.float64).T + ah.astype(np.float64) @ xl.astype(np.float64).T
                                                       + ah.astype(np.float64) @ xh.astype(np.float64).T)
    return slabs


def wgrad_reduce_torch(slabs, Ci, KH, KW):
    """Mirror of conv_wgrad_reduce_torch_kernel (csrc/conv_bwd.hip): the split sum in the reduce kernels' order (S < 16: one running sum;
    else four quarters of eight interleaved running sums, pairwise) and the re-layout packed [n][(tap, c)] -> torch [n][c][kh][kw], channels
    from Ci on dropped."""
    S, N, K = slabs.shape
    Ctot = K // (KH * KW)
    sl = slabs.astype(np.float32)
    if S < 16:
        tot = np.zeros((N, K), np.float32)
        for z in range(S):
            tot = tot + sl[z]
    else:
        qs = []
        for w in range(4):
            z0, z1 = (S * w) // 4, (S * (w + 1)) // 4
            a = [np.zeros((N, K), np.float32) for _ in range(8)]
            z = z0
            while z + 7 < z1:
                for j in range(8):
                    a[j] = a[j] + sl[z + j]
                z += 8
            while z < z1:
                a[0] = a[0] + sl[z]
                z += 1
            qs.append(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])))
        tot = (qs[0] + qs[1]) + (qs[2] + qs[3])
    return np.ascontiguousarray(tot.reshape(N, KH, KW, Ctot)[..., :Ci].transpose(0, 3, 1, 2))


# ---- strip-walker kernels (csrc/conv_strip.hip): LDS patch addressing and the first stage's channel order ----
DS_READ_B128_GROUPS = ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                       [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63])


def strip_px_addr(i, j):
    """byte offset of 16-byte piece j of pixel record i inside the hi part of a plane (px_addr in conv_strip.hip)."""
    return i * 64 + (((j + (i >> 1)) & 3) << 4)


def strip_read_conflict_degree(i0):
    """Worst number of distinct addresses on one 16-byte bank slot among the lanes the LDS serves together, for a ds_read_b128 of
    the A fragment whose first pixel record is i0: lane l reads record i0 + (l & 15), piece l >> 4 (MFMA 16x16x32 operand layout)."""
    worst = 0
    for group in DS_READ_B128_GROUPS:
        slots = {}
        for lane in group:
            a = strip_px_addr(i0 + (lane & 15), lane >> 4)
            slots.setdefault((a // 16) % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def strip_conv1_k_order():
    """Reference input channel (c * 16 + s) held at position k' of a tap's 32-deep reduction in the first-stage strip kernel:
    k' = kg * 8 + sb * 4 + t * 2 + c  <->  s = 4 kg + 2 t + sb."""
    order = []
    for k in range(32):
        kg, sb, t, c = k >> 3, (k >> 2) & 1, (k >> 1) & 1, k & 1
        order.append(c * 16 + 4 * kg + 2 * t + sb)
    return order


# ---- shared-patch engine (csrc/conv_patch.hip): LDS patch addressing (128-byte rows, eight 16-byte pieces) ----
def patch_row_addr(row, piece):
    """byte offset of split32 piece `piece` (0..3 hi, 4..7 lo) of patch row `row`: LDS piece = (piece + (row & 6)) & 7."""
    return (row << 7) | (((piece + (row & 6)) & 7) << 4)


def patch_read_conflict_degree(row0, lo=0, xor_map=False):
    """Worst number of distinct addresses on one 16-byte bank slot among the lanes the LDS serves together, for a ds_read_b128 of
    the pixel fragment whose first patch row is row0 (any shift: taps move the window by one row / one patch line): lane l reads row
    row0 + (l & 15), piece (l >> 4) + 4 lo.  xor_map: the piece ^ ((row >> 1) & 7) map of conv_dma.hip (conflict-free only at
    row0 % 4 == 0, which is all that engine needs)."""
    worst = 0
    for group in DS_READ_B128_GROUPS:
        slots = {}
        for lane in group:
            row, piece = row0 + (lane & 15), (lane >> 4) + 4 * lo
            a = ((row << 7) | ((piece ^ ((row >> 1) & 7)) << 4)) if xor_map else patch_row_addr(row, piece)
            slots.setdefault((a // 16) % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def patch_engine_layer(x, w, transposed, whole):
    """Index arithmetic of csrc/conv_patch.hip on one image (numpy, fp64 sums): x [H, W, C] NHWC input, w the torch weight
    (Conv2d [Co, Ci, 4, 4] / ConvTranspose2d [Ci, Co, 4, 4]), stride 2, pad 1.  For every class (gh, gw) of a conv's window / every
    phase (ph, pw) of a transposed conv the engine stages ONE patch P of the tile's output pixels and its four taps tt = 2a + b
    read it at a row / column shift:
      whole = True   patch row (k, l) = in(2k + 1 - gh, 2l + 1 - gw) (conv) / in(k, l) (transposed); tap (a, b) of output pixel
                     (q, r) reads patch (q + a - 1 + g_h, r + b - 1 + g_w), zeros where that falls off the grid (edge / kill bits)
      whole = False  halo patch P[i][j] = in(2i + gh - 1, 2j + gw - 1) / in(i + ph - 1, j + pw - 1), i <= rows, j <= W (zeros
                     outside the image); tap (a, b) reads P[q + a][r + b]
    with the weight tap (kh, kw) = (2a + gh, 2b + gw) (conv) / (th, tw) = (ph ? a : 1 - a, pw ? b : 1 - b) of the phase's 2x2 window
    (kh = (ph ? 2 : 1) + th (ph ? -2 : 2), as pack_convT_weight).  Returns the NHWC output."""
    H, W, C = x.shape
    if transposed:
        Hq, Wq, Co = H, W, w.shape[1]
        out = np.zeros((2 * H, 2 * W, Co))
    else:
        Hq, Wq, Co = H // 2, W // 2, w.shape[0]
        out = np.zeros((Hq, Wq, Co))

    def inp(i, j):
        return x[i, j].astype(np.float64) if 0 <= i < H and 0 <= j < W else np.zeros(C)

    for g_h in range(2):
        for g_w in range(2):
            if whole:
                P = np.zeros((Hq, Wq, C))
                for k in range(Hq):
                    for l in range(Wq):
                        P[k, l] = inp(k, l) if transposed else inp(2 * k + 1 - g_h, 2 * l + 1 - g_w)
            else:
                P = np.zeros((Hq + 1, Wq + 1, C))
                for i in range(Hq + 1):
                    for j in range(Wq + 1):
                        P[i, j] = inp(i + g_h - 1, j + g_w - 1) if transposed else inp(2 * i + g_h - 1, 2 * j + g_w - 1)
            for a in range(2):
                for b in range(2):
                    if transposed:
                        th, tw = (a if g_h else 1 - a), (b if g_w else 1 - b)
                        kh, kw = (2 if g_h else 1) + th * (-2 if g_h else 2), (2 if g_w else 1) + tw * (-2 if g_w else 2)
                        wt = w[:, :, kh, kw].astype(np.float64)              # [Ci, Co]
                    else:
                        wt = w[:, :, 2 * a + g_h, 2 * b + g_w].astype(np.float64).T   # [Ci, Co]
                    for q in range(Hq):
                        for r in range(Wq):
                            if whole:
                                k, l = q + a - 1 + g_h, r + b - 1 + g_w
                                # the kernel's edge / kill bits: top & (a == 0, g_h == 0), bottom & (a == 1, g_h == 1), likewise left / right
                                killed = (q == 0 and a == 0 and g_h == 0) or (q == Hq - 1 and a == 1 and g_h == 1) or \
                                         (r == 0 and b == 0 and g_w == 0) or (r == Wq - 1 and b == 1 and g_w == 1)
                                assert killed == (not (0 <= k < Hq and 0 <= l < Wq))
                                v = np.zeros(C) if killed else P[k, l]
                            else:
                                v = P[q + a, r + b]
                            if transposed:
                                out[2 * q + g_h, 2 * r + g_w] += v @ wt
                            else:
                                out[q, r] += v @ wt
    return out


def philox_exp1(seed, counter):
    """numpy restatement of csrc/rl_ops.hip philox_exp1 (Philox4x32-10, word 0, 23 bits -> (0, 1) -> -log).  -> (noise, u)."""
    M32 = np.uint64(0xFFFFFFFF)
    counter = np.asarray(counter, dtype=np.uint64)
    c0, c1 = counter & M32, counter >> np.uint64(32)
    c2, c3 = np.full_like(c0, 0x6d32685f), np.zeros_like(c0)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0, n1 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32, p1 & M32
        n2, n3 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32, p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M32, (k1 + np.uint64(0xBB67AE85)) & M32
    u = ((c0 >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    return -np.log(u.astype(np.float32)), u
