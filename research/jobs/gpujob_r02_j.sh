mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "read_" 2>&1 | tail -2
timeout 300 python tools/kbench.py 2>/dev/null | head -9
