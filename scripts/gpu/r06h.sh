#!/bin/bash
cd /root/repo
timeout 300 python scripts/dev/tcol_blocks.py 2>&1 | grep -v amdgpu.ids
