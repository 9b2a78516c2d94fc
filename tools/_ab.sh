mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/ab_tests.log 2>&1; tail -2 gpurun_out/ab_tests.log
python3 tools/train_probe.py C5 2e5 1 8 batched 2>&1 | grep "^C"
python3 tools/train_probe.py C5 2000 5 100 batched 2>&1 | grep "^C"
for c in C3 C4 C5; do python3 tools/wide_steps_time.py $c 2>&1 | grep "G=3\|G=6"; done
