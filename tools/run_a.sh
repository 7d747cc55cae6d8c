mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t_full.log 2>&1
rc=$?
tail -6 gpurun_out/t_full.log
exit $rc
