#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python3 tools/probe/r05_stream_queue_probe.py 2>&1 | grep -v amdgpu.ids; done
echo "== GPU_MAX_HW_QUEUES=8"
for i in 1 2 3; do GPU_MAX_HW_QUEUES=8 python3 tools/probe/r05_stream_queue_probe.py 2>&1 | grep -v amdgpu.ids; done
