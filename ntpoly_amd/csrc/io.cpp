// MatrixMarket / binary IO (PSMatrixModule.F90:351-745, distributed_includes/WriteToMatrixMarket.f90,
// distributed_includes/WriteMatrixToBinary.f90:19-65, MatrixMarketModule.F90).  The reference uses
// MPI-IO; here the root rank reads or writes the whole file (setup path) and the matrix is
// distributed by the triplet fill.
#include "io.hpp"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <fstream>
#include <sstream>

namespace ntp {

void read_matrix_market_file(const std::string& path, HostTriplets& t, int* rows, int* cols, int want_complex, bool header_only) {
  std::ifstream f(path);
  if (!f) NTP_FATAL("cannot open matrix market file " + path);  // PSMatrixModule.F90:405-415
  std::string line;
  if (!std::getline(f, line)) NTP_FATAL("empty matrix market file " + path);
  std::string lower = line;
  std::transform(lower.begin(), lower.end(), lower.begin(), [](unsigned char c) { return (char)std::tolower(c); });
  std::istringstream hs(lower);
  std::string banner, object, format, field, symmetry;
  hs >> banner >> object >> format >> field >> symmetry;
  if (format != "coordinate") NTP_FATAL("only coordinate MatrixMarket files are supported: " + path);
  const bool file_complex = field == "complex";
  const bool pattern = field == "pattern";
  while (std::getline(f, line))
    if (!line.empty() && line[0] != '%') break;
  long long r = 0, c = 0, nnz = 0;
  {
    std::istringstream ss(line);
    ss >> r >> c >> nnz;
  }
  *rows = (int)r;
  *cols = (int)c;
  const bool out_complex = want_complex < 0 ? file_complex : (want_complex != 0);
  t = HostTriplets();
  t.cplx = out_complex;
  if (r < 1 || c < 1 || nnz < 0) NTP_FATAL("bad size line in matrix market file " + path);
  if (header_only) return;
  t.col.reserve((size_t)nnz * 2);
  t.row.reserve((size_t)nnz * 2);
  auto push = [&](int row, int col, double re, double im) {
    t.col.push_back(col);
    t.row.push_back(row);
    t.val.push_back(re);
    if (out_complex) t.val.push_back(im);
  };
  for (long long i = 0; i < nnz; ++i) {
    int row = 0, col = 0;
    double re = 1.0, im = 0.0;
    f >> row >> col;
    if (!pattern) f >> re;
    if (file_complex) f >> im;
    if (!f) NTP_FATAL("truncated matrix market file " + path);
    if (row < 1 || row > r || col < 1 || col > c)
      NTP_FATAL("entry " + std::to_string(i + 1) + " (" + std::to_string(row) + ", " + std::to_string(col) + ") of " + path +
                " lies outside the " + std::to_string(r) + " x " + std::to_string(c) + " matrix of its header");
    push(row, col, re, im);
    if (row != col) {
      if (symmetry == "symmetric") push(col, row, re, im);
      else if (symmetry == "skew-symmetric") push(col, row, -re, -im);
      else if (symmetry == "hermitian") push(col, row, re, -im);
    }
  }
}

void ps_read_matrix_market(PSMatrix& m, const std::string& path, const ProcessGrid* g) {
  int rows = 0, cols = 0;
  HostTriplets t;
  // every rank parses the header (cheap), only the root reads and contributes the entries
  read_matrix_market_file(path, t, &rows, &cols, -1, world().rank != 0);
  ps_construct_empty(m, rows, g, t.cplx);
  ps_fill_from_triplets(m, t);
}

void ps_write_matrix_market(const PSMatrix& m, const std::string& path) {
  use_grid_comm(m.grid);
  const int64_t total = ps_size(m);
  HostTriplets t;
  if (world().active()) {
    DevMat full = ps_gather_full(m);
    to_triplets(full, 0, t);
  } else {
    ps_get_triplets(m, t);
  }
  if (world().rank == 0) {
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) NTP_FATAL("cannot open " + path + " for writing");
    std::fprintf(f, "%%%%MatrixMarket matrix coordinate %s general\n%%\n", m.cplx ? "complex" : "real");
    std::fprintf(f, "%d %d %lld\n", m.dim, m.dim, (long long)total);
    const size_t n = t.size();
    for (size_t i = 0; i < n; ++i) {
      if (m.cplx) std::fprintf(f, "%d %d %.17g %.17g\n", t.row[i], t.col[i], t.val[2 * i], t.val[2 * i + 1]);
      else std::fprintf(f, "%d %d %.17g\n", t.row[i], t.col[i], t.val[i]);
    }
    std::fclose(f);
  }
  comm_barrier();
}

// header int32[3] {rows, cols, is_complex}, int64 total, then {int32 col, int32 row, f64 val
// (| f64 re, f64 im)} in native endianness (WriteMatrixToBinary.f90:43-65)
void ps_write_binary(const PSMatrix& m, const std::string& path) {
  use_grid_comm(m.grid);
  const int64_t total = ps_size(m);
  HostTriplets t;
  if (world().active()) {
    DevMat full = ps_gather_full(m);
    to_triplets(full, 0, t);
  } else {
    ps_get_triplets(m, t);
  }
  if (world().rank == 0) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) NTP_FATAL("cannot open " + path + " for writing");
    int32_t header[3] = {m.dim, m.dim, m.cplx ? 1 : 0};
    std::fwrite(header, sizeof(int32_t), 3, f);
    std::fwrite(&total, sizeof(int64_t), 1, f);
    const size_t n = t.size(), w = m.cplx ? 2 : 1;
    for (size_t i = 0; i < n; ++i) {
      std::fwrite(&t.col[i], sizeof(int32_t), 1, f);
      std::fwrite(&t.row[i], sizeof(int32_t), 1, f);
      std::fwrite(&t.val[i * w], sizeof(double), w, f);
    }
    std::fclose(f);
  }
  comm_barrier();
}

void ps_read_binary(PSMatrix& m, const std::string& path, const ProcessGrid* g) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) NTP_FATAL("cannot open binary matrix file " + path);
  int32_t header[3];
  int64_t total = 0;
  if (std::fread(header, sizeof(int32_t), 3, f) != 3 || std::fread(&total, sizeof(int64_t), 1, f) != 1)
    NTP_FATAL("truncated binary matrix file " + path);
  const bool z = header[2] != 0;
  HostTriplets t;
  t.cplx = z;
  if (world().rank == 0) {
    const size_t w = z ? 2 : 1;
    t.col.resize((size_t)total);
    t.row.resize((size_t)total);
    t.val.resize((size_t)total * w);
    for (int64_t i = 0; i < total; ++i) {
      if (std::fread(&t.col[(size_t)i], sizeof(int32_t), 1, f) != 1 || std::fread(&t.row[(size_t)i], sizeof(int32_t), 1, f) != 1 ||
          std::fread(&t.val[(size_t)i * w], sizeof(double), w, f) != w)
        NTP_FATAL("truncated binary matrix file " + path);
      if (t.col[(size_t)i] < 1 || t.col[(size_t)i] > header[1] || t.row[(size_t)i] < 1 || t.row[(size_t)i] > header[0])
        NTP_FATAL("entry " + std::to_string(i + 1) + " of " + path + " lies outside the matrix of its header");
    }
  }
  std::fclose(f);
  ps_construct_empty(m, header[0], g, z);
  ps_fill_from_triplets(m, t);
}

}  // namespace ntp
