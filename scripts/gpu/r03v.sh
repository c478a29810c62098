#!/bin/bash
cd /root/repo

bash scripts/dev/ab.sh r03v NSVD_LIB_PATH=/root/repo/scripts/_diag/libnsvd_exp_NOPM.so 2>&1 | grep -v "^$" | tail -20
