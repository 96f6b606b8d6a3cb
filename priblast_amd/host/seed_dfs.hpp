// Seed search over the query suffix array x page suffix array (reference: SeedSearch::Run /
// SeedSearchCore, seed_search.cpp:30-45, 153-230): depth-first enumeration of the six ordered
// base-pair types, narrowing one SA interval on each side per character; a branch is emitted
// (and cut) the first time its stacking energy drops below the -e threshold with at least
// `delta` pairs.  The enumeration is tiny (a few thousand nodes per query); the expansion of
// the emitted interval pairs into seed hits - where the volume is - runs on the GPU.
#pragma once
#include <cstdint>
#include <vector>

#include "db_format.hpp"
#include "energy.hpp"

namespace prb {

struct SeedCandidate { // Hit_candidate, hit.hpp:120-146
  int32_t sp_q, ep_q;   // query SA interval
  int32_t sp_db, ep_db; // page SA interval
  int32_t length;
  int32_t query;        // index in the batch (set by the caller)
  double score;         // stacking energy, kcal/mol
};

void seed_dfs(const EnergyParams &p, const uint8_t *qenc, int32_t qn, const int32_t *qsa, const DbPage &page,
              int hash_size, int max_seed_length, int min_accessible_length, double hybrid_threshold,
              std::vector<SeedCandidate> &out);

} // namespace prb
