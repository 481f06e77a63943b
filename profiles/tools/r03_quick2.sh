#!/bin/bash
echo "== 50 then 100 (step trace)"; NB_TWO_STAGE_TRACE=1 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -v amdgpu.ids | grep "^D=\|ms per call"
echo "== 100 (step trace)"; NB_TWO_STAGE_TRACE=1 python profiles/tools/accept_bench.py 100 2>&1 | grep -v amdgpu.ids | grep "^D=\|ms per call"
