timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "xcc or adam" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_trainer_gpu.py -q -x -m gpu -k "ragged or fused_front" 2>&1 | tail -3
