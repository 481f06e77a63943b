set -x
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 | tee gpurun_out/r04/suite_final.log
bash profiles/tools/r04_bench_prof.sh 2>&1 | tail -25
python profiles/tools/accept_bench.py 50 100 2>/dev/null | tee gpurun_out/r04/accept_bench_50_then_100_one_process.txt
python profiles/tools/train_speed.py 2>/dev/null | tee gpurun_out/r04/train_speed_final.txt
