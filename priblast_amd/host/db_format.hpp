// On-disk database of the reference, read and written unchanged:
//   .bas  4 ints {hash_size, repeat_flag, maximal_span, min_accessible_length}
//         (db_construction.cpp:423-436; rna_interaction_search_parameters.cpp:97-114)
//   .seq  per page: int nseq, nseq x int length, int nchars, nchars x uchar codes (each
//         sequence reversed + 0)                   (db_construction.cpp:371-392, db_reader.cpp:73-120)
//   .acc  per sequence: int n1 = L-delta+1, n1 floats, int n2 = L, n2 floats
//                                                    (raccess.cpp:447-480, db_reader.cpp:133-150)
//   .nam  one name per line
//   .ind  per page: int n, n x int SA, start_hash levels 0..h-1 (4^(i+1) ints), end_hash likewise
//                                                    (db_construction.cpp:394-421, db_reader.cpp:158-174)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace prb {

struct DbPage {
  int32_t nseq = 0;
  std::vector<int32_t> seq_length;     // as stored
  std::vector<int32_t> start_pos;      // running sum of length+1 (db_reader.cpp:107-112)
  std::vector<int32_t> seq_length_rep; // number of codes 2..5 per sequence (db_reader.cpp:122-131)
  std::vector<uint8_t> seqs;           // page text
  std::vector<int32_t> sa;
  std::vector<std::vector<int32_t>> start_hash, end_hash;
  // accessibilities, padded to L floats per sequence and concatenated: sequence id starts
  // at acc_base[id] = start_pos[id] - id
  std::vector<float> acc, cond;
  std::vector<std::string> names;
  int64_t acc_base(int id) const { return (int64_t)start_pos[id] - id; }
};

struct DbHeader {
  int32_t hash_size = 0, repeat_flag = 0, maximal_span = 0, min_accessible_length = 0;
};

// returns "" or an error message
std::string read_db(const std::string &prefix, DbHeader &hdr, std::vector<DbPage> &pages);

// k-mer lookup table of a page (db_construction.cpp:337-369): SA interval of every string of
// length <= hash_size over {A,C,G,U}; absent k-mers are stored as (1, 0).
void build_kmer_table(const std::vector<uint8_t> &text, const std::vector<int32_t> &sa, int hash_size,
                      std::vector<std::vector<int32_t>> &start_hash, std::vector<std::vector<int32_t>> &end_hash);

// Narrow the suffix-array interval [*s, *e] to the suffixes whose character at `offset` is c
// (SeedSearchNextCharacter, seed_search.cpp:232-295; Search, db_construction.cpp:438-...).
void sa_narrow(const uint8_t *text, const int32_t *sa, int32_t *s, int32_t *e, uint8_t c, int32_t offset);

struct DbWriter {
  std::string prefix;
  std::string open(const std::string &prefix, const DbHeader &hdr); // truncates the five files
  // appends one page (.seq, .ind) and its sequences' accessibilities / names (.acc, .nam)
  std::string append_page(const DbPage &page, int delta);
};

} // namespace prb
