#!/bin/bash
# prints what the GPU box's host gives this job: CPU topology, the cgroup quota and the allowed CPU list
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|L3|MHz|^CPU\(s\)"
echo "allowed: $(grep Cpus_allowed_list /proc/self/status)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
echo "numa of GPU: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
