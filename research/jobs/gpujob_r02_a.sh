bash research/jobs/gpujob_parity.sh r02_a
timeout 600 python -m pytest tests -m gpu -x -q -k "ignore or id_assign or lstt_forward" > gpurun_out/r02_a_pytest_quick.log 2>&1; tail -3 gpurun_out/r02_a_pytest_quick.log
timeout 400 python bench.py > gpurun_out/r02_a_bench_x3.json 2>gpurun_out/r02_a_bench.err; cut -c1-700 gpurun_out/r02_a_bench_x3.json
timeout 300 python tools/kbench.py > gpurun_out/r02_a_kbench.json 2>gpurun_out/r02_a_kbench.err; cat gpurun_out/r02_a_kbench.json | cut -c1-1500
