#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03s; mkdir -p $O
export TMPDIR=/tmp
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/hwmon_ls.txt 2>&1
timeout 300 python tools/power_probe.py > $O/power_probe.json 2> $O/power_probe.err
