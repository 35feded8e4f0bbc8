"""How far ahead of the GPU the host runs: time to ENQUEUE one joint step against the time the step takes.
usage: host_bound.py B [frame size] [f32|bf16]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import loans_amd
from loans_amd.datasets import synthetic
from loans_amd.runtime import training
B = int(sys.argv[1]); hw, crop = (int(sys.argv[2]) if len(sys.argv) > 2 else 224), 75
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f32'
from loans_amd import ops
if os.environ.get('LOANS_TUNE_FILE') is None and os.path.exists('profiles/r2_cfg3_tune.json') and dtype == 'bf16':
    ops.load_tune_table('profiles/r2_cfg3_tune.json')
ops.set_compute_dtype(dtype); ops.set_storage_dtype(dtype)
dev = torch.device('cuda', 0)
frames = torch.from_numpy(synthetic.make_frames(1, B, hw, hw)).to(dev)
real, labels = synthetic.make_assessor_batch(2, B, crop, crop)
real, labels = torch.from_numpy(real).to(dev), torch.from_numpy(labels).to(dev)
np.random.seed(0)
loc, dis = loans_amd.SheepLocalizer((crop, crop)), loans_amd.ResnetAssessor()
loc.param_predictor.W.set_logical((1e-3 * np.random.standard_normal((6, 512))).astype(np.float32))
with loans_amd.using_config('enable_backprop', False):
    dis(real[:2])
og = loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(loc); od = loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(dis)
upd = loans_amd.SheepAssessor(models=[loc, dis], iterator={'main': training.DeviceBatchIterator([frames]), 'real': training.DeviceBatchIterator([(real, labels)])},
                              optimizer={'opt_gen': og, 'opt_dis': od}, converter=training.identity_converter, device=0)
for _ in range(4): upd.update()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    t0 = time.perf_counter(); upd.update(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print('B=%d %dpx %s: host enqueue %.2f ms, step %.2f ms' % (B, hw, dtype, np.median(enq) * 1e3, np.median(tot) * 1e3))
