#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( time python3 -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/pytest_call31.log 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -25 gpurun_out/pytest_call31.log
