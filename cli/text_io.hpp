// cli/text_io.hpp -- the text writers of the drop-in tools: writeData(file, data, delim[, precision]) of
// util/text_io.hxx:103-133 as hmt/main_merge_order_pb.cxx:37-38 and hmt/main_merge_order_bc.cxx:148-157 call it.  Every element is
// followed by the delimiter; a precision <= 0 keeps the stream's default (6 significant digits).  No HIP dependency: the
// CPU suite builds cli/text_io_check from this header and compares its bytes with the reference's own writers
// (oracle/_ref/ref_misc, tests/test_oracle_vs_ref.py).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

namespace cli {

[[noreturn]] inline void perr(const std::string& msg) {   // glia_base.hxx:66-69
  std::cerr << msg << std::endl;
  exit(EXIT_FAILURE);
}

inline void writeOrder(const std::string& file, const std::vector<uint32_t>& o, int64_t n) {
  std::ofstream os(file);
  if (!os) perr("Error: cannot create file " + file);
  for (int64_t i = 0; i < n; ++i) os << o[3 * i] << " " << o[3 * i + 1] << " " << o[3 * i + 2] << "\n";
}
inline void writeDoubles(const std::string& file, const double* d, int64_t n, int precision = -1) {
  std::ofstream os(file);
  if (!os) perr("Error: cannot create file " + file);
  if (precision > 0) os.precision(precision);
  for (int64_t i = 0; i < n; ++i) os << d[i] << "\n";
}
inline void writeRows(const std::string& file, const double* d, int64_t rows, int cols, int precision) {
  std::ofstream os(file);
  if (!os) perr("Error: cannot create file " + file);
  if (precision > 0) os.precision(precision);
  for (int64_t i = 0; i < rows; ++i) { for (int k = 0; k < cols; ++k) os << d[i * cols + k] << " "; os << "\n"; }
}

}  // namespace cli
