import os, subprocess, sys
ROOT = '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd()
CHILD = r'''
import sys, time, torch, ctypes
sys.path.insert(0, %r)
from nautilus_amd import emulator
def fit(x, y, e, n_epochs):
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n_epochs, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); return time.perf_counter() - t
d, n_row, e = 50, 24000, 4
x = torch.randn((n_row, d), dtype=torch.float64, device='cuda'); y = torch.rand(n_row, dtype=torch.float64, device='cuda')
fit(x, y, e, 2)
res = []
for rep in range(4):
    t_s, t_l = fit(x, y, e, 32), fit(x, y, e, 96)
    res.append((t_l - t_s) / (64 * 120) * 1e6)
print(' '.join('%%.2f' %% r for r in res), 'x=%%x' %% x.data_ptr())
''' % ROOT
for tag in sys.argv[1:]:
    for r in range(5):
        lib = os.path.join(ROOT, 'nautilus_amd', 'lib', 'libnautilus_hip%s.so' % ('' if tag == 'current' else '_' + tag))
        p = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, NAUTILUS_HIP_LIB=lib), capture_output=True, text=True)
        print(tag, p.stdout.strip() or p.stderr[-200:], flush=True)
