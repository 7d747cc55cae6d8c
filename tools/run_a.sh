mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_packed_chunks or hit_cap or concurrent_workers or probe_on_the_chunk" > gpurun_out/t.log 2>&1
rc=$?
tail -40 gpurun_out/t.log
exit $rc
