# in-kernel cycle breakdown of nb_eval_fast_kernel (wavefront 0 of workgroup 0)
#   make debug DEFS=-DNB_DBG_TIMING
#   NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_dbg.so python profiles/tools/fast_ts.py [D]
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch, time
from oracle import mlp_oracle as mo
from nautilus_amd import device, _lib
d = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(d)
A = rng.normal(size=(d, d)); cov = A @ A.T / d + np.eye(d); B = np.linalg.cholesky(cov * 0.02)
nets = [mo.glorot_init(d, e)[:2] for e in range(4)]
ell = device.member(0.5 * np.ones(d), B)
nbd = device.DeviceBound(d, [ell], None, False, [dict(ellipsoid=ell, score_predict_min=0.0, mlp=dict(mean=np.zeros(d), scale=np.ones(d), nets=nets))])
n = 1 << 20
torch.manual_seed(0)
x = torch.rand((n, d), dtype=torch.float64, device='cuda')
for _ in range(3): nbd.neural_score(x)
torch.cuda.synchronize()
buf = torch.zeros(32, dtype=torch.int64, device='cuda')
lib = _lib.load()
lib.nb_set_eval_counters(buf.data_ptr())
t = time.perf_counter(); nbd.neural_score(x); torch.cuda.synchronize(); dt = time.perf_counter() - t
lib.nb_set_eval_counters(None)
c = buf.cpu().numpy()[8:15]
names = ['points+cube+ellipsoid', 'census+standardise', 'stage 1 compute (x E)', 'stage 1 flush+wait+barrier',
         'stage 2 compute (x E)', 'stage 2 flush+wait+barrier', 'epilogue']
tot = c.sum()
print('launch %.3f ms, %d ticks (100 MHz clock64 -> %.3f ms)' % (dt * 1e3, tot, tot / 1e5))
for nm, v in zip(names, c): print('%-30s %10d  %5.1f%%' % (nm, v, 100 * v / tot))
