// FASTA input with the reference's conventions (fastafile_reader.cpp:373-409): the first line
// is a header, a name is the whole header line minus '>', sequence lines are concatenated and
// trailing CR/LF are stripped.
#pragma once
#include <string>
#include <vector>

namespace prb {
// returns "" or an error message
std::string read_fasta(const std::string &path, std::vector<std::string> &names, std::vector<std::string> &seqs);
} // namespace prb
