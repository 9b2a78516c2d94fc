set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "batched or fused_training or training_vs" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
python tools/train_probe.py C2 1e6 1 20 batched 2>&1 | grep "^C" > gpurun_out/tp.log
python tools/train_probe.py C2 2000 5 50 batched 2>&1 | grep "^C" >> gpurun_out/tp.log
python tools/train_probe.py C2 10000 1 50 batched 2>&1 | grep "^C" >> gpurun_out/tp.log
cat gpurun_out/tp.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/tools/train_probe.py C2 1e6 1 10 batched > /dev/null 2>&1
