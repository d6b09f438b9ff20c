cd "${GRAFT_REPO_ROOT:-/root/repo}"
# round 6, final: every configuration, kernel trace and counters of the final tree (research/jobs/gpujob_profile_r06.sh), then the GPU suite
bash research/jobs/gpujob_profile_r06.sh r06 > gpurun_out/r06_profile_job.log 2>&1
tail -25 gpurun_out/r06_profile_job.log
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06/pytest_gpu.log
