#include "fasta.hpp"

#include <fstream>

namespace prb {

std::string read_fasta(const std::string &path, std::vector<std::string> &names, std::vector<std::string> &seqs) {
  std::ifstream fp(path);
  if (!fp) return "Error: can't open input_file:" + path;
  names.clear();
  seqs.clear();
  std::string line, cur;
  bool first = true;
  while (std::getline(fp, line)) {
    if (first || (!line.empty() && line[0] == '>')) {
      if (!first) seqs.push_back(cur);
      names.push_back(line.empty() ? line : line.substr(1));
      cur.clear();
      first = false;
    } else {
      while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
      cur += line;
    }
  }
  if (!first) seqs.push_back(cur);
  return "";
}

} // namespace prb
