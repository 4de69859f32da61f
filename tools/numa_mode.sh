#!/bin/bash
# Runs ON THE GPU BOX: tools/numa_mode_probe.py unpinned and pinned to every NUMA node, several processes each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
NODES=$(ls -d /sys/devices/system/node/node[0-9]* | sed 's/.*node//' | sort -n | tr '\n' ' ')
echo "nodes: $NODES"
for rep in 1 2 3; do
  for node in none $NODES; do
    timeout -k 10 120 python $R/tools/numa_mode_probe.py $node 2>/dev/null | grep '^{"pinned'
  done
done
echo "== kernel arguments forced into device memory (HIP_FORCE_DEV_KERNARG=1) / host memory (=0), unpinned"
for rep in 1 2 3 4 5 6; do
  for v in 1 0; do
    HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python $R/tools/numa_mode_probe.py none 2>/dev/null | grep '^{"pinned'
  done
done
