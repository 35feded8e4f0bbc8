import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import loans_amd
from loans_amd.iou.iou_regressor import BottleneckB
from loans_amd.sheep.resnet import BasicA
from loans_amd.runtime.core import Variable
from oracle import model as M
from oracle.model import _ResUnit
from tests.gpu_util import dev

def run(kind, compute, storage):
    rng = np.random.RandomState(3); np.random.seed(4)
    B, H, W = 8, 12, 10
    w = loans_amd.links.HeNormal()
    if kind == 'basic_a':
        cin, cout = 64, 128
        blk = BasicA(cout, 2, in_ch=cin); stages = [('conv1','bn1',2,1),('conv2','bn2',1,1)]; sc = ('conv3','bn3',2,1)
    else:
        cin = cout = 128
        blk = BottleneckB(cout, 32, w); stages = [('conv1','bn1',1,0),('conv2','bn2',1,1),('conv3','bn3',1,0)]; sc = None
    for key, p in blk.namedparams():
        if key.endswith('/gamma'): p.set_logical((1 + 0.2 * rng.standard_normal(p.logical_shape)).astype(np.float32))
        elif key.endswith('/beta'): p.set_logical((0.2 * rng.standard_normal(p.logical_shape)).astype(np.float32))
    blk.finalize(torch.device('cuda', 0))
    lp = M.cast_params(blk.state_dict_chainer(), np.float64)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32)).to(torch.bfloat16).float().numpy()
    xt = dev(np.ascontiguousarray(x.transpose(0, 2, 3, 1)))
    loans_amd.set_compute_dtype(compute)
    if storage == 'bf16': loans_amd.set_storage_dtype('bf16'); xt = xt.to(torch.bfloat16)
    xv = Variable(xt, requires_grad=True)
    out = blk(xv)
    unit = _ResUnit(lp, stages, sc, True)
    o_out = unit.fwd(x.astype(np.float64))
    gy = torch.from_numpy(rng.standard_normal(o_out.shape).astype(np.float32)).to(torch.bfloat16).float().numpy()
    g = dev(np.ascontiguousarray(gy.transpose(0, 2, 3, 1)))
    out.grad = g.to(out.data.dtype)
    blk.cleargrads(); out.backward()
    grads = {}
    gx_ref = unit.bwd(gy.astype(np.float64), grads)
    gx = xv.grad.float().cpu().numpy().transpose(0, 3, 1, 2)
    l2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    worst = max(l2(p.grad_logical(), grads[k[1:]]) for k, p in blk.namedparams())
    print('%-10s compute %-4s storage %-4s | out l2 %.2e | gx l2 %.3e frac>3%%max %.4f | worst param-grad l2 %.3e' % (
        kind, compute, storage, l2(out.data.float().cpu().numpy().transpose(0, 3, 1, 2), o_out), l2(gx, gx_ref),
        (np.abs(gx - gx_ref) > 0.03 * np.abs(gx_ref).max()).mean(), worst))
    loans_amd.set_compute_dtype('f32')

for kind in ('basic_a', 'chainer_b'):
    for c, s in (('f32', 'f32'), ('bf16', 'f32'), ('bf16', 'bf16')):
        run(kind, c, s)
