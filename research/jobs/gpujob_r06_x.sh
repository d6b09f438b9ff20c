cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call x: mid-clip new objects inside batched clips (BatchedDeAOTEngine.add_reference_slots)
O=$PWD/gpurun_out/r06x; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_batched.py -q -m gpu -x -s -k "mid_clip or queue_refill or slots_in_different" 2>&1 | tail -30 | tee $O/pytest.txt
