cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tools/probe/rambo_traffic 1000000 40 > gpurun_out/rambo_traffic.txt 2>&1; cat gpurun_out/rambo_traffic.txt
python tools/rambo_time.py > gpurun_out/rambo_time.txt 2>&1; tail -5 gpurun_out/rambo_time.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/full.log 2>&1; tail -5 gpurun_out/full.log
