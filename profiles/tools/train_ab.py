"""Same-box comparison of trainer libraries: us per Adam step (differenced) of
D=50/E=4 and D=100/E=8, every library in its own process, several rounds.
  python profiles/tools/train_ab.py [--rounds 3] lib_tag ... (tags of
  nautilus_amd/lib/libnautilus_hip_<tag>.so; 'current' = the shipped library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from nautilus_amd import emulator
def fit(x, y, e, n_epochs):
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n_epochs, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); return time.perf_counter() - t
out = []
for d, n_row, e in [(50, 24000, 4), (100, 24000, 8)]:
    x = torch.randn((n_row, d), dtype=torch.float64, device='cuda'); y = torch.rand(n_row, dtype=torch.float64, device='cuda')
    fit(x, y, e, 2); fit(x, y, e, 32)
    best = 1e9
    for rep in range(3):
        t_s, t_l = fit(x, y, e, 32), fit(x, y, e, 96)
        best = min(best, (t_l - t_s) / (64 * 120) * 1e6)
    out.append('%%.2f' %% best)
print(' '.join(out))
''' % ROOT
args = sys.argv[1:]
rounds = 3
if args and args[0] == '--rounds':
    rounds = int(args[1]); args = args[2:]
print('us per step, best of 3 differenced fits: D=50/E=4  D=100/E=8')
for r in range(rounds):
    for tag in args:
        lib = os.path.join(ROOT, 'nautilus_amd', 'lib', 'libnautilus_hip%s.so' % (
            '' if tag == 'current' else '_' + tag))
        env = dict(os.environ, NAUTILUS_HIP_LIB=lib)
        p = subprocess.run([sys.executable, '-c', CHILD], env=env,
                           capture_output=True, text=True)
        print('round %d  %-8s %s' % (r, tag, p.stdout.strip() or p.stderr[-300:]),
              flush=True)
