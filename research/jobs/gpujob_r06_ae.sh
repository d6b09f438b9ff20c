cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, call ae: clips in flight through the clip driver (InFlightClipDriver)
O=$PWD/gpurun_out/r06ae; mkdir -p $O
timeout 1500 python -m pytest tests/test_driver.py -q -m gpu -x -k "in_flight or golden or grows_past" 2>&1 | tail -8 | tee $O/pytest.txt
timeout 900 python tools/clips_in_flight_probe.py r50_aotl 6 24 2>&1 | grep -v amdgpu | tee $O/in_flight_aot.txt
timeout 900 python tools/clips_in_flight_probe.py r50_deaotl 6 24 2>&1 | grep -v amdgpu | tee $O/in_flight_deaot.txt
