#!/bin/bash
# round 4, first contact: fp64 arbitration test, RCCL world-1 test, graph-event probe, parity attribution, suite, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -x -s -k "test_480p_lstt_isolated_from_miopen" > $O/lstt_isolated.log 2>&1; tail -30 $O/lstt_isolated.log
timeout 600 python -m pytest tests/test_driver.py -q -m gpu -x -s -k "rccl_world1" > $O/rccl_world1.log 2>&1; tail -5 $O/rccl_world1.log
timeout 300 python tools/graph_event_probe.py > $O/graph_event_probe.json 2> $O/graph_event_probe.err; cat $O/graph_event_probe.json; tail -3 $O/graph_event_probe.err
timeout 900 python tests/probes/parity_attribution.py --tag r04a --out $O/r04a_parity_attribution.json > $O/parity.log 2>&1; tail -c 1500 $O/parity.log
timeout 600 python bench.py > $O/r04a_bench_x3.json 2> $O/bench_x3.err; tail -c 600 $O/r04a_bench_x3.json
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
