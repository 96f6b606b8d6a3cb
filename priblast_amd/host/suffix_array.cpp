#include "suffix_array.hpp"

#include <algorithm>
#include <cstring>

namespace prb {

namespace {

// One level of SA-IS over text s[0..n) with symbols in [0, K).  `sa` has n entries.
// Position n is a virtual sentinel smaller than every symbol: s[n-1] is therefore L-type,
// and the (virtual) LMS suffix at n is the smallest suffix; it is never stored.
template <class Sym> struct Level {
  const Sym *s;
  int32_t *sa;
  int32_t n, K;
  std::vector<uint8_t> stype; // 1 = S-type, 0 = L-type
  std::vector<int32_t> bkt;

  bool is_lms(int32_t i) const { return i > 0 && stype[i] && !stype[i - 1]; }

  void bucket_bounds(bool ends) {
    std::fill(bkt.begin(), bkt.end(), 0);
    for (int32_t i = 0; i < n; i++) bkt[s[i]]++;
    int32_t sum = 0;
    for (int32_t c = 0; c < K; c++) {
      sum += bkt[c];
      bkt[c] = ends ? sum : sum - bkt[c];
    }
  }

  void induce() {
    // L-type suffixes, left to right; the sentinel suffix comes first and induces n-1
    bucket_bounds(false);
    sa[bkt[s[n - 1]]++] = n - 1;
    for (int32_t i = 0; i < n; i++) {
      int32_t j = sa[i];
      if (j > 0 && !stype[j - 1]) sa[bkt[s[j - 1]]++] = j - 1;
    }
    // S-type suffixes, right to left
    bucket_bounds(true);
    for (int32_t i = n - 1; i >= 0; i--) {
      int32_t j = sa[i];
      if (j > 0 && stype[j - 1]) sa[--bkt[s[j - 1]]] = j - 1;
    }
  }

  // LMS substrings starting at a and b (a != b): equal in symbols, types and length?
  bool lms_equal(int32_t a, int32_t b) const {
    for (int32_t d = 0;; d++) {
      if (a + d >= n || b + d >= n) return false; // one of them runs into the sentinel
      if (s[a + d] != s[b + d] || stype[a + d] != stype[b + d]) return false;
      if (d > 0) {
        bool ea = is_lms(a + d), eb = is_lms(b + d);
        if (ea || eb) return ea && eb;
      }
    }
  }

  void run() {
    if (n == 1) {
      sa[0] = 0;
      return;
    }
    stype.assign(n, 0);
    for (int32_t i = n - 2; i >= 0; i--)
      stype[i] = s[i] < s[i + 1] || (s[i] == s[i + 1] && stype[i + 1]);
    bkt.assign(K, 0);

    // step 1: sort the LMS substrings by one round of induced sorting
    bucket_bounds(true);
    std::fill(sa, sa + n, -1);
    for (int32_t i = 1; i < n; i++)
      if (is_lms(i)) sa[--bkt[s[i]]] = i;
    induce();

    // compact the sorted LMS positions to the front
    int32_t n1 = 0;
    for (int32_t i = 0; i < n; i++)
      if (is_lms(sa[i])) sa[n1++] = sa[i];
    if (n1 == 0) return; // no LMS suffix besides the sentinel: induce() already sorted everything

    // step 2: name the LMS substrings (names go to sa[n1 + pos/2])
    std::fill(sa + n1, sa + n, -1);
    int32_t names = 0, prev = -1;
    for (int32_t i = 0; i < n1; i++) {
      int32_t pos = sa[i];
      if (prev < 0 || !lms_equal(prev, pos)) names++;
      prev = pos;
      sa[n1 + pos / 2] = names - 1;
    }
    // reduced text, in text order, at the tail of sa
    for (int32_t i = n - 1, j = n - 1; i >= n1; i--)
      if (sa[i] >= 0) sa[j--] = sa[i];
    int32_t *s1 = sa + n - n1, *sa1 = sa;

    // step 3: suffix array of the reduced text
    if (names < n1) {
      Level<int32_t> sub;
      sub.s = s1;
      sub.sa = sa1;
      sub.n = n1;
      sub.K = names;
      sub.run();
    } else {
      for (int32_t i = 0; i < n1; i++) sa1[s1[i]] = i;
    }

    // step 4: map back to text positions and induce the final order
    for (int32_t i = 1, j = 0; i < n; i++)
      if (is_lms(i)) s1[j++] = i;
    for (int32_t i = 0; i < n1; i++) sa1[i] = s1[sa1[i]];
    std::fill(sa + n1, sa + n, -1);
    bucket_bounds(true);
    for (int32_t i = n1 - 1; i >= 0; i--) {
      int32_t j = sa[i];
      sa[i] = -1;
      sa[--bkt[s[j]]] = j;
    }
    induce();
  }
};

} // namespace

void suffix_array(const uint8_t *text, int32_t n, int32_t *sa) {
  if (n <= 0) return;
  Level<uint8_t> top;
  top.s = text;
  top.sa = sa;
  top.n = n;
  top.K = 256;
  top.run();
}

} // namespace prb
