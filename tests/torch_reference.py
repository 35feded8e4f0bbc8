"""Independent torch-CPU composition of the LoANs graph (autograd does the
backward).  Third leg of the oracle: used ONLY to cross-check the
NumPy restatement in oracle/, never by the product."""
import numpy as np
import torch
import torch.nn.functional as F

STAGES = (('res2', 1), ('res3', 2), ('res4', 2), ('res5', 2))
MEAN = [103.063, 115.903, 123.152]


def to_torch(params, dtype):
    out = {}
    for k, v in params.items():
        if v.dtype.kind == 'f':
            t = torch.tensor(v, dtype=dtype)
            if not k.endswith(('/avg_mean', '/avg_var')):
                t.requires_grad_(True)
            out[k] = t
    return out


def _cbn(p, x, conv, bn, stride, pad, train):
    c = F.conv2d(x, p[conv + '/W'], p.get(conv + '/b'), stride=stride, padding=pad)
    return F.batch_norm(c, p[bn + '/avg_mean'].clone(), p[bn + '/avg_var'].clone(), p[bn + '/gamma'], p[bn + '/beta'],
                        training=train, momentum=0.1, eps=2e-5)


def _block(p, x, prefix, stride, train):
    a = prefix + '/0'
    h1 = F.relu(_cbn(p, x, a + '/conv1', a + '/bn1', stride, 1, train))
    h1 = _cbn(p, h1, a + '/conv2', a + '/bn2', 1, 1, train)
    h2 = _cbn(p, x, a + '/conv3', a + '/bn3', stride, 1, train)
    x = F.relu(h1 + h2)
    b = prefix + '/1'
    h = F.relu(_cbn(p, x, b + '/conv1', b + '/bn1', 1, 1, train))
    h = _cbn(p, h, b + '/conv2', b + '/bn2', 1, 1, train)
    return F.relu(h + x)


def localizer(p, images, out_size, train=True):
    dtype = images.dtype
    x = (images.float() * 255).to(torch.uint8).to(dtype)
    x = (x.float().flip(1) - torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)).to(dtype)   # chainer: float32 arithmetic
    fe = 'feature_extractor'
    h = F.relu(_cbn(p, x, fe + '/conv1', fe + '/bn1', 2, 3, train))
    h = F.max_pool2d(h, 3, 2, ceil_mode=True)
    names = [(fe + '/' + n, s) for n, s in STAGES]
    H = images.shape[-2]
    if H > 224:
        names.append(('res6', 2))
        if H > 300:
            names.append(('res7', 2))
    for prefix, stride in names:
        h = _block(p, h, prefix, stride, train)
    h = h.mean(dim=(2, 3))
    theta = F.linear(h, p['param_predictor/W'], p['param_predictor/b']).view(-1, 2, 3)
    mask = torch.ones(2, 3, dtype=dtype); mask[0, 1] = 0; mask[1, 0] = 0
    theta = theta * mask
    grid = F.affine_grid(theta, (len(images), 3) + tuple(out_size), align_corners=True)
    rois = F.grid_sample(images, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    return rois, grid.permute(0, 3, 1, 2), theta


def assessor(p, x):
    def c(k, t, s, pad):
        return F.conv2d(t, p[k + '/W'], None, stride=s, padding=pad)
    h = c('r0/c1', F.relu(c('r0/c0', x, 1, 1)), 2, 1) + c('r0/cs', x, 2, 1)
    h = c('r1/c1', F.relu(c('r1/c0', F.relu(h), 1, 1)), 2, 1) + c('r1/cs', h, 2, 1)
    h = c('r2/c1', F.relu(c('r2/c0', F.relu(h), 1, 1)), 1, 1) + h
    h = c('r3/c1', F.relu(c('r3/c0', F.relu(h), 1, 1)), 1, 1) + h
    return torch.sigmoid(F.linear(F.relu(h).flatten(1), p['l4/W']))


def regularisers(points, image_hw):
    # (torch's relu / clamp give an exact tie NO gradient, Chainer's F.maximum(x, zeros) gives it to x: this composition
    # cross-checks the oracle away from ties only; the tie convention has its own known-answer tests,
    # tests/test_oracle_kat.py::test_kat9_ties_follow_chainers_maximum_and_absolute)
    H, W = image_hw
    th, tw = points.shape[-2:]
    g = (points + 1) / 2
    xs, ys = g[:, 0] * W, g[:, 1] * H
    direction = F.relu(ys[:, 0, 0] - ys[:, th - 1, 0]).mean() + F.relu(xs[:, 0, 0] - xs[:, 0, tw - 1]).mean()
    bbox = torch.cat([points[:, 0, 0, 0], points[:, 1, 0, 0], points[:, 0, 0, tw - 1], points[:, 1, th - 1, 0]])
    out = torch.clamp(bbox + 1, max=0).abs().sum() + torch.clamp(bbox - 1, min=0).sum()
    return direction, out


def step_losses(lp, dp, frames, real, labels, out_size, target=1.0):
    rois, points, theta = localizer(lp, frames, out_size)
    y_fake = assessor(dp, rois)
    loss_loc = F.mse_loss(y_fake, torch.full_like(y_fake, target))
    d, o = regularisers(points, frames.shape[-2:])
    loss_loc = loss_loc + d + o
    y_real = assessor(dp, real)
    loss_dis = F.mse_loss(y_real, labels)
    return loss_loc, loss_dis, dict(rois=rois, points=points, theta=theta, y_fake=y_fake, y_real=y_real)


# ---- ResNet-50 localizer (Resnet50SheepLocalizer, sheep_localizer.py:120-178) ----
R50 = (('res2', 3, 1), ('res3', 4, 2), ('res4', 6, 2), ('res5', 3, 2))


def _unit(p, x, stages, shortcut, train):
    h = x
    for i, (c, b, s, pad) in enumerate(stages):
        h = _cbn(p, h, c, b, s, pad, train)
        if i < len(stages) - 1:
            h = F.relu(h)
    sc = x if shortcut is None else _cbn(p, x, shortcut[0], shortcut[1], shortcut[2], shortcut[3], train)
    return F.relu(h + sc)


def localizer50(p, images, out_size, train=True):
    dtype = images.dtype
    x = (images.float() * 255).to(torch.uint8).float()
    x = (x.flip(1) - torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)).to(dtype)
    fe = 'feature_extractor'
    h = F.relu(_cbn(p, x, fe + '/conv1', fe + '/bn1', 2, 3, train))
    h = F.max_pool2d(h, 3, 2, ceil_mode=True)
    for name, n, stride in R50:
        a = '%s/%s/a' % (fe, name)
        h = _unit(p, h, [(a + '/conv1', a + '/bn1', stride, 0), (a + '/conv2', a + '/bn2', 1, 1), (a + '/conv3', a + '/bn3', 1, 0)],
                  (a + '/conv4', a + '/bn4', stride, 0), train)
        for i in range(1, n):
            b = '%s/%s/b%d' % (fe, name, i)
            h = _unit(p, h, [(b + '/conv1', b + '/bn1', 1, 0), (b + '/conv2', b + '/bn2', 1, 1), (b + '/conv3', b + '/bn3', 1, 0)], None, train)
    H = images.shape[-2]
    for name in (['res6'] if H > 224 else []) + (['res7'] if H > 300 else []):
        for blk in ('a', 'b1'):
            q = '%s/%s' % (name, blk)
            s = 2 if blk == 'a' else 1
            sc = (q + '/residual_conv/conv', q + '/residual_conv/bn', 2, 0) if blk == 'a' else None
            h = _unit(p, h, [(q + '/conv1/conv', q + '/conv1/bn', 1, 0), (q + '/conv2/conv', q + '/conv2/bn', s, 1),
                             (q + '/conv3/conv', q + '/conv3/bn', 1, 0)], sc, train)
    h = h.mean(dim=(2, 3))
    theta = F.linear(h, p['param_predictor/W'], p['param_predictor/b']).view(-1, 2, 3)
    mask = torch.ones(2, 3, dtype=dtype); mask[0, 1] = 0; mask[1, 0] = 0
    theta = theta * mask
    grid = F.affine_grid(theta, (len(images), 3) + tuple(out_size), align_corners=True)
    rois = F.grid_sample(images, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    return rois, grid.permute(0, 3, 1, 2), theta
