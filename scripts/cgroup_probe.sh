#!/bin/bash
# Is the process being throttled by the container's CPU quota?  Prints cpu.max and the throttling counters around a run.
show() { for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpu/cpu.stat; do [ -r $f ] && { echo "== $f"; cat $f; }; done; }
nproc; show
"$@"
show
