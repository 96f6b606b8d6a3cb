// Suffix array construction by induced sorting (SA-IS, Nong, Zhang & Chan 2009), written
// for this project.  Replaces the reference's third-party sais (sais.cpp:656-667, used at
// rna_interaction_search.cpp:252-262 for queries and db_construction.cpp:331-335 for pages).
// Convention: plain byte/int text, no unique terminator required; a suffix that is a prefix
// of another sorts first (virtual sentinel) - the same order the reference's SA has.
#pragma once
#include <cstdint>
#include <vector>

namespace prb {

// SA[0..n) <- suffix array of text[0..n); alphabet 0..255
void suffix_array(const uint8_t *text, int32_t n, int32_t *sa);

} // namespace prb
