#!/bin/bash
# Round 5, fifth session: the whole GPU suite and the default bench line on
# the session's tree (blobs for device likelihoods / sharded runs, no host
# path in the mixture fit, one wait per batch less).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -35 > $O/suite.log
tail -30 $O/suite.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
