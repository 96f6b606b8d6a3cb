#include "seed_dfs.hpp"

namespace prb {

namespace {

// the six (query code, db code) pairs in the reference's order: CG GC GU UG AU UA
// (seed_search.hpp:38-50)
constexpr int kPairs[6][2] = {{3, 4}, {4, 3}, {4, 5}, {5, 4}, {2, 5}, {5, 2}};

struct Dfs {
  const EnergyParams &p;
  const uint8_t *q;
  const int32_t *qsa;
  const DbPage &pg;
  int hash_size, max_len, delta;
  double thr;
  std::vector<SeedCandidate> &out;
  int qpath[64], dpath[64];

  void visit(int32_t spq, int32_t epq, int32_t spd, int32_t epd, double score, int depth) {
    if (depth >= max_len) return;
    int32_t iv[6][4];
    for (int c = 0; c < 6; c++) {
      int32_t s = spq, e = epq;
      sa_narrow(q, qsa, &s, &e, (uint8_t)kPairs[c][0], depth);
      iv[c][0] = s;
      iv[c][1] = e;
      if (depth + 1 > hash_size) {
        s = spd;
        e = epd;
        sa_narrow(pg.seqs.data(), pg.sa.data(), &s, &e, (uint8_t)kPairs[c][1], depth);
      } else { // k-mer table: base-4 number of the db path (seed_search.cpp:183-193)
        int32_t code = kPairs[c][1] - 2;
        for (int j = 0; j < depth; j++) code += (dpath[j] - 2) << (2 * (depth - j));
        s = pg.start_hash[depth][code];
        e = pg.end_hash[depth][code];
      }
      iv[c][2] = s;
      iv[c][3] = e;
    }
    for (int c = 0; c < 6; c++) {
      if (iv[c][0] > iv[c][1] || iv[c][2] > iv[c][3]) continue;
      double s2 = 0.0;
      if (depth > 0) { // stacking of the new pair on the previous one (seed_search.cpp:203-208)
        const int type = p.bp_pair[qpath[depth - 1] - 1][dpath[depth - 1] - 1];
        const int type2 = p.rtype[p.bp_pair[kPairs[c][0] - 1][kPairs[c][1] - 1]];
        s2 = score + ((double)p.stack37[type][type2]) / 100;
      }
      if (s2 < thr && depth + 1 >= delta) {
        out.push_back({iv[c][0], iv[c][1], iv[c][2], iv[c][3], depth + 1, 0, s2});
      } else {
        qpath[depth] = kPairs[c][0];
        dpath[depth] = kPairs[c][1];
        visit(iv[c][0], iv[c][1], iv[c][2], iv[c][3], s2, depth + 1);
      }
    }
  }
};

} // namespace

void seed_dfs(const EnergyParams &p, const uint8_t *qenc, int32_t qn, const int32_t *qsa, const DbPage &page,
              int hash_size, int max_seed_length, int min_accessible_length, double hybrid_threshold,
              std::vector<SeedCandidate> &out) {
  if (qn <= 0 || page.sa.empty()) return;
  if (max_seed_length > 63) max_seed_length = 63;
  Dfs d{p, qenc, qsa, page, hash_size, max_seed_length, min_accessible_length, hybrid_threshold, out, {}, {}};
  d.visit(0, qn - 1, 0, (int32_t)page.sa.size() - 1, 0.0, 0);
}

} // namespace prb
