mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --config 720p_k8 --steps 20 --warmup 5 --gap 2 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_720p.json; cut -c1-1200 gpurun_out/bench_720p.json
python tools/kbench.py --h 46 --w 81 --cap 8 --iters 10 2>/dev/null | tail -30 > gpurun_out/kbench_720p.json; cat gpurun_out/kbench_720p.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1c -o r1c -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
ls -la gpurun_out/prof_r1c gpurun_out/pmc_fetch gpurun_out/pmc_write
