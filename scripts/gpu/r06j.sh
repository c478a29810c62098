#!/bin/bash
cd /root/repo
timeout 600 python scripts/dev/cdk_diffmap.py 2>&1 | grep -v amdgpu.ids | tail -24
