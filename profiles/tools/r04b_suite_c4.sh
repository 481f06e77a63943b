mkdir -p gpurun_out/r04b_final
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 > gpurun_out/r04b_final/suite.log; tail -3 gpurun_out/r04b_final/suite.log
timeout 900 python examples/run_config.py C4 --counters > gpurun_out/r04b_final/c4_full.json 2> gpurun_out/r04b_final/c4_full.err; cut -c1-400 gpurun_out/r04b_final/c4_full.json
