// Internal declarations shared by the host (g++) and device (hipcc) halves of libspada_spgemm.so.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "spada_ffi.h"

// Library-owned host CSR in the reference's CsrMatStorage layout (storage.rs:150-160).
struct spada_host_csr {
    uint64_t rows = 0, cols = 0;
    std::vector<uint64_t> indptr;   // rows + 1
    std::vector<uint64_t> indices;  // nnz
    std::vector<double> data;       // nnz
    uint64_t nnz() const { return indptr.empty() ? 0 : indptr.back(); }
};

namespace spada {
// Records the message returned by spada_last_error() and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();
// SPADA_TRACE=0/1/2: lines on stderr at or below the level set in the environment (replaces util.rs:1-24)
int trace_level();
void trace(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline spada_csr_view view_of(const spada_host_csr &m)
{
    return spada_csr_view{m.rows, m.cols, m.nnz(), m.indptr.data(), m.indices.data(), m.data.data()};
}
}  // namespace spada
