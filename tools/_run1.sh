cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err; tail -c 300 gpurun_out/bench_r03b.json; tail -3 gpurun_out/bench_r03b.err
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_C4 -o k -- python3 $R/tools/run_config_step.py C4 6 > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_C5 -o k -- python3 $R/tools/run_config_step.py C5 6 > /dev/null 2>&1 )
ls gpurun_out/trace_C4
