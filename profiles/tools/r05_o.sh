#!/bin/bash
export TMPDIR=/tmp
python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_sampler_gpu.py -q -m gpu -x -k "gmm or sharded or rccl or split" 2>&1 | tail -4
NB_GMM_MAX_WGS=1 timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "gmm" 2>&1 | tail -2
