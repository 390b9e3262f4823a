#!/bin/bash
# K fresh processes of tools/population_roles.py on one box.  usage: bash tools/population_roles.sh [K] [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-8}; mkdir -p $R/gpurun_out
for i in $(seq 1 $K); do python $R/tools/population_roles.py ${2:-3} 2>/dev/null ; done | tee $R/gpurun_out/population_roles.txt
