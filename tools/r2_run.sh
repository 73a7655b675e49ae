#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/r2_probe.py 2>&1 | grep -v amdgpu
