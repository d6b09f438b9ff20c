#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_driver.py -q -m gpu -k "contract" 2>&1 | tail -5
