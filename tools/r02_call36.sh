#!/bin/bash
mkdir -p gpurun_out
QUDA_AMD_INVERT_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call36.log 2>&1; echo rc=$?; grep -i "invert profile\|invertQuda\|phase\|h2d\|d2h\|load\|save" gpurun_out/c5_call36.log | tail -24 | cut -c1-300
