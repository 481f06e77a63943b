#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('value %.4g ms_per_step %.3f full %.4g setup %.2f' % (r['value'], r['ms_per_step'], r['value_full_run'], r['setup_s']))"; done
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_sampler_behaviour_gpu.py tests/test_io_gpu.py tests/test_live_gpu.py -q -m gpu -x 2>&1 | tail -3
