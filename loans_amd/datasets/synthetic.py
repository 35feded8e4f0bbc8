"""Seeded synthetic paste-and-crop inputs.

Mirrors the recipe of the reference's offline generator
(datasets/sheep/paste_and_crop_sheep.py:109-136 paste, :13,45-81 IoU-targeted
crop) without PIL: a low-frequency uint8 background, an RGBA "stamp" blob of
random size in [H/15,H/2]x[W/15,W/2] alpha-composited at a uniform random
position; frames are ``uint8/255`` as float32 CHW RGB exactly like
common/datasets/image_dataset.py:98 produces them, so the localizer's uint8
truncation (sheep_localizer.py:45,72-82) is lossless on them.

Pure NumPy; no device code.  Used by bench.py, the tests and the CPU oracle.
"""
import numpy as np

IOU_RANGES = [v / 100.0 for v in range(20, 105, 5)]   # paste_and_crop_sheep.py:13


def _low_freq_noise(rng, h, w, cells=8):
    coarse = rng.integers(0, 256, size=(cells + 1, cells + 1, 3)).astype(np.float32)
    ys = np.linspace(0, cells, h, dtype=np.float32)
    xs = np.linspace(0, cells, w, dtype=np.float32)
    y0 = np.minimum(ys.astype(np.int32), cells - 1)
    x0 = np.minimum(xs.astype(np.int32), cells - 1)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a = coarse[y0][:, x0]
    b = coarse[y0][:, x0 + 1]
    c = coarse[y0 + 1][:, x0]
    d = coarse[y0 + 1][:, x0 + 1]
    img = (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx)
    img += rng.normal(0, 6.0, size=img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def _stamp(rng, sh, sw):
    yy, xx = np.mgrid[0:sh, 0:sw].astype(np.float32)
    cy, cx = (sh - 1) / 2.0, (sw - 1) / 2.0
    r = np.sqrt(((yy - cy) / max(cy, 1)) ** 2 + ((xx - cx) / max(cx, 1)) ** 2)
    alpha = np.clip((1.05 - r) * 6.0, 0, 1)[..., None]
    colour = rng.integers(0, 256, size=3).astype(np.float32)
    tex = colour[None, None, :] + rng.normal(0, 12.0, size=(sh, sw, 3))
    return np.clip(tex, 0, 255).astype(np.float32), alpha.astype(np.float32)


def make_composite(rng, h, w):
    """Returns (uint8 HWC image, paste box (x0, y0, x1, y1))."""
    bg = _low_freq_noise(rng, h, w).astype(np.float32)
    sw = int(rng.integers(max(w // 15, 1), max(w // 2, 2) + 1))
    sh = int(rng.integers(max(h // 15, 1), max(h // 2, 2) + 1))
    px = int(rng.integers(0, w - sw + 1))
    py = int(rng.integers(0, h - sh + 1))
    tex, alpha = _stamp(rng, sh, sw)
    region = bg[py:py + sh, px:px + sw]
    bg[py:py + sh, px:px + sw] = region * (1 - alpha) + tex * alpha
    return np.clip(bg + 0.5, 0, 255).astype(np.uint8), (px, py, px + sw, py + sh)


def to_chw_float(img_u8):
    """image_dataset.py:98 : uint8 HWC -> float32 CHW in [0,1] (exactly k/255)."""
    return (img_u8.astype(np.float32) / np.float32(255)).transpose(2, 0, 1)


def make_frames(seed, batch, h, w):
    """The ``main`` iterator's batch: (B,3,H,W) float32 RGB in [0,1]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.stack([to_chw_float(make_composite(rng, h, w)[0]) for _ in range(batch)], axis=0)


def _iou(a, b):
    ix = max(0, min(a[2], b[2]) - max(a[0], b[0]))
    iy = max(0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = ix * iy
    ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / ua if ua > 0 else 0.0


def _resize_bilinear(img, th, tw):
    h, w, _ = img.shape
    ys = (np.arange(th, dtype=np.float32) + 0.5) * h / th - 0.5
    xs = (np.arange(tw, dtype=np.float32) + 0.5) * w / tw - 0.5
    ys = np.clip(ys, 0, h - 1)
    xs = np.clip(xs, 0, w - 1)
    y0 = np.minimum(ys.astype(np.int32), max(h - 2, 0)); y1 = np.minimum(y0 + 1, h - 1)
    x0 = np.minimum(xs.astype(np.int32), max(w - 2, 0)); x1 = np.minimum(x0 + 1, w - 1)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    f = img.astype(np.float32)
    out = (f[y0][:, x0] * (1 - fy) * (1 - fx) + f[y0][:, x1] * (1 - fy) * fx +
           f[y1][:, x0] * fy * (1 - fx) + f[y1][:, x1] * fy * fx)
    return np.clip(out + 0.5, 0, 255).astype(np.uint8)


def make_assessor_batch(seed, batch, th, tw, src=224):
    """The ``real`` iterator's batch: crops around the paste box with target IoU
    cycling 0.20..1.00 (paste_and_crop_sheep.py:45-48), resized to (th,tw);
    labels = IoU(crop, paste) rounded to 4 decimals (:221-222), shape (B,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    images, labels = [], []
    for i in range(batch):
        img, box = make_composite(rng, src, src)
        desired = min(IOU_RANGES[i % len(IOU_RANGES)], 1.0)
        bw, bh = box[2] - box[0], box[3] - box[1]
        best, best_iou = box, 1.0
        for _ in range(200):
            dev = 1.0 - desired
            cw = int(rng.integers(max(int(bw - bw * dev), 1), int(bw + bw * dev) + 1))
            ch = int(rng.integers(max(int(bh - bh * dev), 1), int(bh + bh * dev) + 1))
            if desired < 0.3:
                cw = int(min(bw + dev * 10 * bw, src)); ch = int(min(bh + dev * 10 * bh, src))
            dx = int(cw // 2 * dev); dy = int(ch // 2 * dev)
            lo_x, hi_x = max(box[0] - dx, 0), max(min(box[0] + dx, src - cw), 0)
            lo_y, hi_y = max(box[1] - dy, 0), max(min(box[1] + dy, src - ch), 0)
            cx = int(rng.integers(min(lo_x, hi_x), max(lo_x, hi_x) + 1))
            cy = int(rng.integers(min(lo_y, hi_y), max(lo_y, hi_y) + 1))
            crop = (cx, cy, min(cx + cw, src), min(cy + ch, src))
            iou = _iou(crop, box)
            if abs(iou - desired) < abs(best_iou - desired):
                best, best_iou = crop, iou
            if desired - 0.05 < iou <= desired:
                break
        crop_img = img[best[1]:max(best[3], best[1] + 1), best[0]:max(best[2], best[0] + 1)]
        images.append(to_chw_float(_resize_bilinear(crop_img, th, tw)))
        labels.append(round(float(best_iou), 4))
    return np.stack(images, axis=0), np.asarray(labels, np.float32).reshape(-1, 1)
