// CPUs this process may actually keep busy: the hardware threads, cut down by the affinity mask and by the
// cgroup's CPU bandwidth limit (cpu.max of cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us of v1).  The pools of
// host threads (suffix arrays, seed DFS, result lines) are sized from it: a container with 256 visible hardware
// threads and a quota of 16 CPUs throttles EVERY thread of the process for the rest of a 100 ms period once 32
// busy threads have used the quota up - the thread that feeds the GPU included (measured on the GPU box: the search
// stream idle for 30-200 ms at a time, 8 % of a configs[2] step).
#pragma once

namespace prb {

int cpu_budget();
// default size of one pool of host threads (PRB_HOST_THREADS overrides): half the budget, at most 32 - two pools
// are busy at a time (the seed DFS of the coming queries, the result lines of the last batch)
int default_host_threads();

} // namespace prb
