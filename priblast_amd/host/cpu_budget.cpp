#include "cpu_budget.hpp"

#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace prb {

namespace {

// quota / period of a "<quota|max> <period>" file (cgroup v2 cpu.max); 0 = no limit
double read_cpu_max(const char *path) {
  FILE *f = std::fopen(path, "r");
  if (!f) return 0;
  char q[64] = {0};
  double period = 0;
  const int n = std::fscanf(f, "%63s %lf", q, &period);
  std::fclose(f);
  if (n != 2 || std::strcmp(q, "max") == 0 || period <= 0) return 0;
  const double quota = std::atof(q);
  return quota > 0 ? quota / period : 0;
}

double read_number(const char *path) {
  FILE *f = std::fopen(path, "r");
  if (!f) return 0;
  double v = 0;
  const int n = std::fscanf(f, "%lf", &v);
  std::fclose(f);
  return n == 1 ? v : 0;
}

} // namespace

int cpu_budget() {
  static const int budget = [] {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    double limit = read_cpu_max("/sys/fs/cgroup/cpu.max");
    if (limit <= 0) {
      const double quota = read_number("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), period = read_number("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
      if (quota > 0 && period > 0) limit = quota / period;
    }
    if (limit > 0) n = std::min(n, std::max(1, (int)std::ceil(limit)));
    return n;
  }();
  return budget;
}

int default_host_threads() { return std::max(1, std::min(32, cpu_budget() / 2)); }

} // namespace prb
