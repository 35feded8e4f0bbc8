"""The imgaug branch of ``ImageDataset`` (reference common/datasets/image_dataset.py:57-70,80-83):

    iaa.Sometimes(p, iaa.SomeOf((0, None), [iaa.Fliplr(1.0),
                                            iaa.AddToHueAndSaturation(iap.Uniform(-20, 20), per_channel=True),
                                            iaa.CropAndPad(percent=(-0.10, 0.10), pad_mode=["constant", "edge"])],
                                random_order=True))

imgaug (and the OpenCV it calls) cannot be installed here, so the three operations are RESTATED from their documented
behaviour -- with probability p, a random subset (0 to all) of them in random order: mirror; add independent U(-20, 20)
samples to the hue and saturation channels of the uint8 HSV image (H in [0, 180), the addition clips to [0, 255], the hue wraps
on the way back); crop (negative) or pad (positive, constant 0 or edge fill) every side by an independent U(-10 %, 10 %) of the
frame and resize back to the frame's size (bilinear, 11-bit fixed-point weights).  The random streams and OpenCV's exact
roundings are not reproducible: this branch is PARITY-UNPINNED by construction (augmentation noise); what is pinned is that the
host form here (NumPy) and the GPU form (csrc/augment.hip, one launch per position of the order) give the same bytes.
"""
import numpy as np

OPS = ('flip', 'huesat', 'croppad')


def sample_params(rng, H, W, probability):
    """what one image gets: a list of up to three 8-int parameter rows, one per position of the sampled order
    (row[0]: 0 none, 1 flip, 2 hue / saturation, 3 crop-and-pad); ``rng`` is a ``random.Random``"""
    rows = []
    if rng.random() < probability:                               # Sometimes(p, ...)
        n = rng.randint(0, len(OPS))                             # SomeOf((0, None), ...): how many
        order = rng.sample(range(len(OPS)), n)                   # which, in random order
        for op in order:
            if op == 0:
                rows.append([1, 0, 0, 0, 0, 0, 0, 0])
            elif op == 1:
                rows.append([2, int(round(rng.uniform(-20, 20))), int(round(rng.uniform(-20, 20))), 0, 0, 0, 0, 0])
            else:
                t, r, b, l = (rng.uniform(-0.10, 0.10) for _ in range(4))      # sample_independently: one draw per side
                px = [int(round(t * H)), int(round(r * W)), int(round(b * H)), int(round(l * W))]
                # never crop a frame away: keep at least one row / column
                if H + px[0] + px[2] < 1:
                    px[0] = px[2] = 0
                if W + px[1] + px[3] < 1:
                    px[1] = px[3] = 0
                rows.append([3] + px + [rng.randint(0, 1), 0, 0])
    while len(rows) < len(OPS):
        rows.append([0] * 8)
    return rows


# ---- NumPy form of csrc/augment.hip (same integer arithmetic) --------------------------------------------------------------
def _rgb2hsv(img):
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(r, np.maximum(g, b))
    diff = v - np.minimum(r, np.minimum(g, b))
    s = np.where(v > 0, (255 * diff + v // 2) // np.maximum(v, 1), 0)
    num = np.where(v == r, g - b, np.where(v == g, (b - r) + 2 * diff, (r - g) + 4 * diff))
    d = np.maximum(diff, 1)
    q = 60 * num + np.where(num >= 0, d, -d)
    hh = np.sign(q) * (np.abs(q) // (2 * d))                     # C division truncates toward zero
    hh = np.where(hh < 0, hh + 180, hh)
    h = np.where(diff == 0, 0, np.where(hh >= 180, hh - 180, hh))
    return h, s, v


def _hsv2rgb(h, s, v):
    h = h % 180
    sec, fr = h // 30, h % 30
    p = (v * (255 - s) + 127) // 255
    q = (v * (7650 - s * fr) + 3825) // 7650
    t = (v * (7650 - s * (30 - fr)) + 3825) // 7650
    table = [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)]
    out = np.zeros(h.shape + (3,), np.int64)
    for i, rgb in enumerate(table):
        m = sec == i
        for c in range(3):
            out[..., c] = np.where(m, rgb[c], out[..., c])
    return out.astype(np.uint8)


def _croppad(img, top, right, bottom, left, edge):
    H, W, _ = img.shape
    VH, VW = H + top + bottom, W + left + right

    def coord(n_out, n_in):
        o = np.arange(n_out, dtype=np.int64)
        num = (2 * o + 1) * n_in - n_out
        i0 = np.floor_divide(num, 2 * n_out)
        fr = ((num - i0 * 2 * n_out) * 2048) // (2 * n_out)
        return i0, fr
    y0, fy = coord(H, VH)
    x0, fx = coord(W, VW)

    def fetch(vy, vx):
        vy, vx = np.clip(vy, 0, VH - 1), np.clip(vx, 0, VW - 1)
        sy, sx = vy - top, vx - left
        inside = ((sy >= 0) & (sy < H))[:, None] & ((sx >= 0) & (sx < W))[None, :]
        px = img[np.clip(sy, 0, H - 1)][:, np.clip(sx, 0, W - 1)].astype(np.int64)
        return px if edge else np.where(inside[..., None], px, 0)
    a, b, c, d = fetch(y0, x0), fetch(y0, x0 + 1), fetch(y0 + 1, x0), fetch(y0 + 1, x0 + 1)
    fxx, fyy = fx[None, :, None], fy[:, None, None]
    t0, t1 = a * (2048 - fxx) + b * fxx, c * (2048 - fxx) + d * fxx
    return ((t0 * (2048 - fyy) + t1 * fyy + (1 << 21)) >> 22).astype(np.uint8)


def apply_host(img_u8_hwc, rows):
    """the sampled operations of one image, in order, on a uint8 HWC RGB frame"""
    img = np.ascontiguousarray(img_u8_hwc)
    for row in rows:
        if row[0] == 1:
            img = np.ascontiguousarray(img[:, ::-1])
        elif row[0] == 2:
            h, s, v = _rgb2hsv(img)
            img = _hsv2rgb(np.clip(h + row[1], 0, 255), np.clip(s + row[2], 0, 255), v)
        elif row[0] == 3:
            img = _croppad(img, row[1], row[2], row[3], row[4], row[5])
    return img


def apply_device(frames_u8, rows_per_image):
    """frames_u8: device uint8 [B][H][W][3]; rows_per_image: B lists of three parameter rows.  One launch per position of
    the order that any image uses; returns a new tensor."""
    import ctypes as C
    import torch
    from ... import _lib
    from ...ops import _ptr, _stream, check
    B, H, W, _ = frames_u8.shape
    table = np.asarray(rows_per_image, np.int32).reshape(B, len(OPS), 8)
    cur = frames_u8
    for k in range(len(OPS)):
        if not table[:, k, 0].any():
            continue
        params = torch.from_numpy(np.ascontiguousarray(table[:, k])).to(frames_u8.device)
        out = torch.empty_like(frames_u8)
        check(_lib.load().loans_augment_stage_u8(_ptr(cur), _ptr(out), B, H, W, _ptr(params), _stream()), 'loans_augment_stage_u8')
        cur = out
    return cur
