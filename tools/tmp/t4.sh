mkdir -p gpurun_out/s2
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 | tee gpurun_out/s2/pytest_tail.log
