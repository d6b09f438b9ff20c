mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "read_ or blocked" > gpurun_out/r02_b_pytest.log 2>&1; tail -15 gpurun_out/r02_b_pytest.log
timeout 300 python tools/kbench_read.py > gpurun_out/r02_b_kbench_read.json 2>gpurun_out/r02_b_kbench_read.err; cat gpurun_out/r02_b_kbench_read.json; tail -3 gpurun_out/r02_b_kbench_read.err
