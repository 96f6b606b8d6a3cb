/* Exhaustive check of the division-free z / 100.0 used by the device code
 * (priblast_amd/csrc/search_device.hpp: div100): prints the number of mismatches. */
#include <math.h>
#include <stdio.h>
int main(void) {
  const double r100 = 0.01;
  long bad = 0;
  for (int z = -100000; z <= 100000; z++) {
    const double zd = (double)z, q0 = zd * r100;
    const double r = fma(-q0, 100.0, zd);
    const double q1 = fma(r, r100, q0);
    if (q1 != zd / 100.0 || signbit(q1) != signbit(zd / 100.0)) bad++;
  }
  printf("%ld\n", bad);
  return bad != 0;
}
