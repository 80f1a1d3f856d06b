#!/bin/bash
# what the GPU box gives the CPU baseline: cores, NUMA nodes, cgroup CPU quota, memory
echo "nproc: $(nproc)  online: $(getconf _NPROCESSORS_ONLN)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo n/a)   cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null || echo n/a)"
echo "cfs quota (v1): $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null || echo n/a) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null || echo n/a)"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|L3"
ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l
for n in /sys/devices/system/node/node*; do echo "$n: cpus $(cat $n/cpulist)  $(grep MemTotal $n/meminfo | awk '{print $4, $5}')"; done 2>/dev/null
grep -E "MemTotal|MemAvailable|HugePages_Total|AnonHugePages" /proc/meminfo
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
uptime
