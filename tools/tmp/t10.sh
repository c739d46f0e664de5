timeout 900 python -m pytest tests/test_trainer_gpu.py -q -x -m gpu -k "tail_windows" 2>&1 | grep -v Warning | tail -15
