// On-disk formats of the reference: MatrixMarket text and NTPoly's binary triplet
// format.  Host-side: files are host data; matrices go to the GPU through the triplet fill.
#pragma once
#include <string>

#include "engine.hpp"

namespace ntp {
// parse a MatrixMarket coordinate file into 1-based triplets (symmetric / skew-symmetric /
// hermitian files are expanded like TripletListModule.F90 SymmetrizeTripletList).
// want_complex: convert to that scalar type; -1 keeps the file's type.  Entries outside the header's shape are fatal.
// header_only: stop after the size line (t comes back empty with the scalar type set)
void read_matrix_market_file(const std::string& path, HostTriplets& t, int* rows, int* cols, int want_complex,
                             bool header_only = false);
void ps_read_matrix_market(PSMatrix& m, const std::string& path, const ProcessGrid* g);
void ps_write_matrix_market(const PSMatrix& m, const std::string& path);
void ps_read_binary(PSMatrix& m, const std::string& path, const ProcessGrid* g);
void ps_write_binary(const PSMatrix& m, const std::string& path);
}  // namespace ntp
