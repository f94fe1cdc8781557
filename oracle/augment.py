"""Oracle of the reference's mini-batch augmentation (test infrastructure only).

Restates data/batch_provider.py:140-272 (`_augmentation_function`) for the options the shipped experiments switch on
(phiseg_7_5.py:30-34: rotations, crop-scale; the flips are requested under keys the provider never reads -- SURVEY.md Q6 --
but are restated too) together with the image helpers it calls (reference utils.py:18-38, 86-92):

    rotate_image            = cv2.warpAffine(img, cv2.getRotationMatrix2D((cols/2, rows/2), angle, 1), (cols, rows), INTER_LINEAR)
    rotate_image_as_onehot  = argmax over labels of rotate_image(one-hot float64 planes)
    resize_image            = cv2.resize(crop, (n_y, n_x), INTER_LINEAR)
    resize_image_as_onehot  = argmax over labels of resize_image(one-hot float64 planes)

OpenCV is a third-party dependency that is NOT installed here (parity unpinned at the cv2 level): `warp_affine_linear` and
`resize_linear` below restate OpenCV's published algorithms -- warpAffine's fixed-point coordinate grid (AB_BITS = 10,
INTER_BITS = 5: source coordinates rounded to 1/32 pixel, bilinear weights from the 32-entry table, BORDER_CONSTANT 0) and
resize's half-pixel-centre bilinear with float coefficients -- operation by operation.  The random decisions of the
reference come from the unseeded global numpy RNG; here they are explicit parameters (the product draws them from its
Philox contract, phiseg_code_amd/data/augment.py)."""
import math

import numpy as np

AB_BITS, INTER_BITS = 10, 5
AB_SCALE, INTER_TAB = 1 << AB_BITS, 1 << INTER_BITS


def _cv_round(v):
    """cvRound / saturate_cast<int>(double): round half to even."""
    return np.rint(np.asarray(v, dtype=np.float64)).astype(np.int64)


def rotation_matrix(cols, rows, angle_deg):
    """cv2.getRotationMatrix2D((cols/2, rows/2), angle, 1.0) -> 2x3 float64 (forward map)."""
    cx, cy = cols / 2, rows / 2
    a = math.cos(angle_deg * math.pi / 180.0)
    b = math.sin(angle_deg * math.pi / 180.0)
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)


def invert_affine(M):
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    i00, i01, i10, i11 = A11, -M[0, 1] * D, -M[1, 0] * D, A22
    b1 = -i00 * M[0, 2] - i01 * M[1, 2]
    b2 = -i10 * M[0, 2] - i11 * M[1, 2]
    return np.array([[i00, i01, b1], [i10, i11, b2]], dtype=np.float64)


def warp_grid(M, rows, cols):
    """Fixed-point source coordinates of cv2.warpAffine (without WARP_INVERSE_MAP) -> (sx, sy, a, b) integer arrays [rows, cols]:
    top-left source pixel and the 1/32-pixel fractions."""
    iM = invert_affine(M)
    x = np.arange(cols)
    adelta = _cv_round(iM[0, 0] * x * AB_SCALE)
    bdelta = _cv_round(iM[1, 0] * x * AB_SCALE)
    rd = AB_SCALE // INTER_TAB // 2
    y = np.arange(rows)
    X0 = _cv_round((iM[0, 1] * y + iM[0, 2]) * AB_SCALE) + rd
    Y0 = _cv_round((iM[1, 1] * y + iM[1, 2]) * AB_SCALE) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    return X >> INTER_BITS, Y >> INTER_BITS, X & (INTER_TAB - 1), Y & (INTER_TAB - 1)


def warp_affine_linear(img, M, work=np.float32):
    """cv2.warpAffine(img, M, (cols, rows), flags=INTER_LINEAR), borderMode BORDER_CONSTANT, value 0.
    img [rows, cols] (work=float32: CV_32F) or [rows, cols, C] float64 planes (work=float64: CV_64F); weights are float32."""
    rows, cols = img.shape[:2]
    sx, sy, a, b = warp_grid(M, rows, cols)
    tab = np.arange(INTER_TAB, dtype=np.float32) * np.float32(1.0 / INTER_TAB)
    wx1, wy1 = tab[a], tab[b]
    wx0, wy0 = np.float32(1) - wx1, np.float32(1) - wy1
    w = [(wy0 * wx0), (wy0 * wx1), (wy1 * wx0), (wy1 * wx1)]                 # float32 products, OpenCV's 2-D table

    def tap(dy, dx):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < rows) & (xx >= 0) & (xx < cols)
        v = img[np.clip(yy, 0, rows - 1), np.clip(xx, 0, cols - 1)]
        return np.where(ok if img.ndim == 2 else ok[..., None], v, 0).astype(work)
    ex = (lambda t: t) if img.ndim == 2 else (lambda t: t[..., None])
    out = tap(0, 0) * ex(w[0]).astype(work)
    out = out + tap(0, 1) * ex(w[1]).astype(work)
    out = out + tap(1, 0) * ex(w[2]).astype(work)
    out = out + tap(1, 1) * ex(w[3]).astype(work)
    return out.astype(work)


def _resize_coeffs(src, dst):
    scale = src / float(dst)
    d = np.arange(dst)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0, src - 1
    return s, np.minimum(s + 1, src - 1), (np.float32(1) - f).astype(np.float32), f


def resize_linear(img, out_rows, out_cols, work=np.float32):
    """cv2.resize(img, (out_cols, out_rows), interpolation=INTER_LINEAR): horizontal pass then vertical pass, float32
    coefficients, `work` accumulation (float32 for CV_32F, float64 for CV_64F)."""
    rows, cols = img.shape[:2]
    x0, x1, ax0, ax1 = _resize_coeffs(cols, out_cols)
    y0, y1, by0, by1 = _resize_coeffs(rows, out_rows)
    src = img.astype(work)
    ex = (lambda t: t) if img.ndim == 2 else (lambda t: t[..., None])
    h = src[:, x0] * ex(ax0[None, :]).astype(work) + src[:, x1] * ex(ax1[None, :]).astype(work)      # [rows, out_cols(, C)]
    out = h[y0] * ex(by0[:, None]).astype(work) + h[y1] * ex(by1[:, None]).astype(work)
    return out.astype(work)


def onehot(lbl, nlabels):
    """reference utils.convert_to_onehot (utils.py:86-92): float64 planes."""
    return (lbl[..., None] == np.arange(nlabels)).astype(np.float64)


def augment_pair(img, lbl, p, nlabels):
    """One (image [X, Y] float32, label map [X, Y] uint8) pair through batch_provider.py:186-262 with explicit random decisions
    p = dict(augment, angle, r_y, p_x, p_y, fliplr, flipud) (augment: the coin flip of line 197; fliplr / flipud: lines 249-260)."""
    img = np.asarray(img, dtype=np.float32)
    lbl = np.asarray(lbl, dtype=np.uint8)
    n_x, n_y = img.shape
    if p["augment"]:
        if p.get("angle") is not None:                                        # ROTATE (199-210)
            M = rotation_matrix(n_y, n_x, p["angle"])
            img = warp_affine_linear(img, M, np.float32)
            if nlabels <= 4:
                lbl = np.argmax(warp_affine_linear(onehot(lbl, nlabels), M, np.float64), axis=-1).astype(np.uint8)
            else:
                raise NotImplementedError("more than 4 labels: cv2.INTER_NEAREST branch (not used by the shipped experiments)")
        if p.get("r_y") is not None:                                          # RANDOM CROP SCALE (213-226)
            r, px, py = p["r_y"], p["p_x"], p["p_y"]
            img = resize_linear(img[py:py + r, px:px + r], n_x, n_y, np.float32)
            lbl = np.argmax(resize_linear(onehot(lbl[py:py + r, px:px + r], nlabels), n_x, n_y, np.float64), axis=-1).astype(np.uint8)
    if p.get("fliplr"):
        img, lbl = np.fliplr(img), np.fliplr(lbl)
    if p.get("flipud"):
        img, lbl = np.flipud(img), np.flipud(lbl)
    return np.ascontiguousarray(img), np.ascontiguousarray(lbl)
