cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call ac: augmentation engines of the one-engine-each TTA loop on HIP streams of their own (RMEM_TTA_STREAMS)
O=$PWD/gpurun_out/r06ac; mkdir -p $O
timeout 1500 python -m pytest tests/test_driver.py tests/test_hip_aot.py -q -m gpu -x -k "tta or aug or driver or clip" 2>&1 | tail -8 | tee $O/pytest.txt
timeout 900 python tools/tta_streams_probe.py r50_aotl 24 2>&1 | grep -v amdgpu | tee $O/tta_streams_aot.txt
timeout 900 python tools/tta_streams_probe.py r50_deaotl 24 2>&1 | grep -v amdgpu | tee $O/tta_streams_deaot_serial.txt
