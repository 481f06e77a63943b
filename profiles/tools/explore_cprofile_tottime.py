"""Python-side profile (cProfile, cumulative) of the exploration phase of the
headline run: where the host spends its time (training waits, GMM split,
MVEE batches, read-backs).  python profiles/tools/explore_cprofile.py"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nautilus_amd import GaussianLikelihood, Sampler, unit_prior
d = 50
like = GaussianLikelihood(np.full(d, 0.5), np.eye(d) * 0.05**2)
s = Sampler(unit_prior, like, n_dim=d, n_live=2000, n_networks=4, n_batch=16384, vectorized=True, seed=0)
pr = cProfile.Profile(); pr.enable()
s.run(n_eff=0, n_shell=0, discard_exploration=True, timeout=300)
torch.cuda.synchronize(); pr.disable()
print(s.timing)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
