// C++ mirror of the reference's host-side interface for the SpGEMM path, above the C ABI.
// Same names, argument meaning and result shapes as the Rust originals (citations into
// /root/reference/src); the reference's toolchain (Rust) is absent from the build image, so this is the
// host language the front end is written in.  Errors surface as spada::Error (the reference panics).
//
//   load_mm_mat            py2rust.rs:62-97        parse_config / OmegaConfig   frontend.rs:8-23, :77-85
//   GEMM::from_mat         gemm.rs:41-53           CsrMatStorage::init_with_gemm storage.rs:214-239
//   CsrRow                 storage.rs:34-126       Simulator::{new,execute,get_exec_result,get_*_stat}
//                                                  simulator.rs:431-507, :509-890, :1008-1062
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "spada_ffi.h"

namespace spada {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc)
{
    if (rc != SPADA_OK) throw Error(rc, spada_last_error());
}

// sprs::CsMat<f64> stand-in: shape + the three arrays
struct CsMat {
    uint64_t nrows = 0, ncols = 0;
    std::vector<uint64_t> indptr, indices;
    std::vector<double> data;
    std::pair<uint64_t, uint64_t> shape() const { return {nrows, ncols}; }
    uint64_t rows() const { return nrows; }
    uint64_t cols() const { return ncols; }
    uint64_t nnz() const { return indptr.empty() ? 0 : indptr.back(); }
    spada_csr_view view() const { return spada_csr_view{nrows, ncols, nnz(), indptr.data(), indices.data(), data.data()}; }
    static CsMat take(spada_host_csr *h)
    {
        spada_csr_view v;
        check(spada_host_csr_view(h, &v));
        CsMat m;
        m.nrows = v.rows;
        m.ncols = v.cols;
        m.indptr.assign(v.indptr, v.indptr + v.rows + 1);
        m.indices.assign(v.indices, v.indices + v.nnz);
        m.data.assign(v.data, v.data + v.nnz);
        spada_host_csr_free(h);
        return m;
    }
};

inline CsMat load_mm_mat(const std::string &dir_path, const std::string &gemm_nm)
{
    std::string path = dir_path;
    if (!path.empty() && path.back() != '/') path += '/';
    path += gemm_nm + ".mtx";
    spada_host_csr *h = nullptr;
    check(spada_mtx_read(path.c_str(), &h));
    return CsMat::take(h);
}

using OmegaConfig = spada_config;
inline OmegaConfig parse_config(const std::string &config_fp)
{
    std::printf("%s\n", config_fp.c_str());   // frontend.rs:78
    OmegaConfig c;
    check(spada_config_parse(config_fp.c_str(), &c));
    return c;
}

struct GEMM {
    std::string name;
    std::shared_ptr<CsMat> a, b;   // b == a for square inputs (the reference clones; the values are identical)
    static GEMM from_mat(const std::string &mn, CsMat mat)
    {
        GEMM g;
        g.name = mn;
        g.a = std::make_shared<CsMat>(std::move(mat));
        spada_host_csr *h = nullptr;
        int same = 0;
        spada_csr_view v = g.a->view();
        check(spada_from_mat(&v, &h, &same));
        g.b = same ? g.a : std::make_shared<CsMat>(CsMat::take(h));
        return g;
    }
};

struct CsrRow {
    uint64_t rowptr = 0;
    std::vector<double> data;
    std::vector<uint64_t> indptr;   // column indices (the reference's field name)
    size_t len() const { return indptr.size(); }
    size_t size() const { return data.size() + indptr.size(); }
};

struct CsrMatStorage {
    std::shared_ptr<CsMat> mat;     // data / indptr / indices live here (Vec<f64> / Vec<usize> in the reference)
    uint64_t read_count = 0, write_count = 0;
    uint64_t mat_shape[2] = {0, 0}; // [cols, rows] as in storage.rs:225
    static std::pair<CsrMatStorage, CsrMatStorage> init_with_gemm(const GEMM &gemm)
    {
        CsrMatStorage a, b;
        a.mat = gemm.a;
        a.mat_shape[0] = gemm.a->ncols;
        a.mat_shape[1] = gemm.a->nrows;
        b.mat = gemm.b;
        b.mat_shape[0] = gemm.b->ncols;
        b.mat_shape[1] = gemm.b->nrows;
        return {a, b};
    }
    uint64_t row_num() const { return mat->nrows; }
    const std::vector<uint64_t> &indptr() const { return mat->indptr; }
    // storage.rs:156-157, :252-255.  The order is computed by the device pre-pass of Simulator::execute (key = what
    // sort_by_length / sort_by_products asked for); row_remap holds it afterwards.
    bool remapped = false;
    int reorder_key = SPADA_REORDER_BY_LENGTH;
    std::vector<uint64_t> row_remap;
    void reorder_row(int key)
    {
        remapped = true;
        reorder_key = key;
    }
};
// preprocessing.rs:76-89 (rows of A by ascending length, stable) and its product-aware sibling: both return the key the
// device pre-pass sorts by
inline int sort_by_length(const CsrMatStorage &) { return SPADA_REORDER_BY_LENGTH; }
inline int sort_by_products(const CsrMatStorage &) { return SPADA_REORDER_BY_PRODUCTS; }

enum class Accelerator { Ip, Op, MultiRow, Spada };

class Simulator {
public:
    // Same parameter list as Simulator::new (simulator.rs:431-448).  The accelerator-model parameters are
    // kept for the report; they do not steer the GPU kernels.
    Simulator(uint64_t pe_num, uint64_t at_num, uint64_t lane_num, uint64_t cache_size, uint64_t word_byte,
              uint64_t output_base_addr, const uint64_t default_block_shape[2], CsrMatStorage *a_matrix,
              CsrMatStorage *b_matrix, Accelerator accelerator, uint64_t mem_latency, uint64_t cache_latency, float freq,
              uint64_t channel, float bandwidth_per_channel, int accumulator = SPADA_ACC_LDS_HASH)
        : word_byte_(word_byte ? word_byte : 8), freq_(freq), a_(a_matrix), b_(b_matrix)
    {
        (void)pe_num; (void)at_num; (void)lane_num; (void)cache_size; (void)output_base_addr; (void)default_block_shape;
        (void)accelerator; (void)mem_latency; (void)cache_latency; (void)channel; (void)bandwidth_per_channel;
        spada_options o{sizeof(spada_options), -1, accumulator, 0};
        check(spada_create(&o, &ctx_));
    }
    ~Simulator() { spada_destroy(ctx_); }
    Simulator(const Simulator &) = delete;
    Simulator &operator=(const Simulator &) = delete;

    void execute()
    {
        spada_csr_view va = a_->mat->view(), vb = b_->mat->view();
        uint64_t nnz = 0;
        if (a_->remapped) {   // main.rs:60-63: -p; the product is mapped back by the numeric call (simulator.rs:1039-1055)
            a_->row_remap.assign(va.rows, 0);
            check(spada_spgemm_symbolic_reordered(ctx_, &va, &vb, a_->reorder_key, &nnz, a_->row_remap.data()));
        } else {
            check(spada_spgemm_symbolic(ctx_, &va, &vb, &nnz));
        }
        c_.nrows = va.rows;
        c_.ncols = vb.cols;
        c_.indptr.assign(va.rows + 1, 0);
        c_.indices.assign(nnz, 0);
        c_.data.assign(nnz, 0.0);
        check(spada_spgemm_numeric(ctx_, c_.indptr.data(), c_.indices.data(), c_.data.data()));
        check(spada_get_stats(ctx_, &stats_));
    }
    // one CsrRow per A row, ascending (simulator.rs:1034-1062); `limit` rows are materialised
    std::vector<CsrRow> get_exec_result(size_t limit = SIZE_MAX) const
    {
        std::vector<CsrRow> out;
        for (uint64_t r = 0; r < c_.nrows && out.size() < limit; ++r) {
            CsrRow row;
            row.rowptr = r;
            row.data.assign(c_.data.begin() + c_.indptr[r], c_.data.begin() + c_.indptr[r + 1]);
            row.indptr.assign(c_.indices.begin() + c_.indptr[r], c_.indices.begin() + c_.indptr[r + 1]);
            out.push_back(std::move(row));
        }
        return out;
    }
    const CsMat &result_matrix() const { return c_; }
    const spada_stats &stats() const { return stats_; }
    // The reference's counters are simulated word counts; these are the measured algorithmic traffic of
    // the GPU run in words of `word_byte` (documented re-definition).
    std::pair<uint64_t, uint64_t> get_a_mat_stat() const { return {(stats_.a_nnz * 12 + (stats_.rows + 1) * 8) / word_byte_, 0}; }
    std::pair<uint64_t, uint64_t> get_b_mat_stat() const { return {(stats_.a_nnz * 16 + stats_.nprod * 12) / word_byte_, 0}; }
    std::pair<uint64_t, uint64_t> get_c_mat_stat() const { return {0, stats_.bytes_write / word_byte_}; }
    std::pair<uint64_t, uint64_t> get_cache_stat() const { return {0, 0}; }
    // kernel time x freq (GHz) = cycles of the simulated clock
    uint64_t get_exec_cycle() const { return (uint64_t)((stats_.ms_symbolic_call + stats_.ms_numeric_call) * 1e6 * freq_); }

private:
    uint64_t word_byte_;
    float freq_;
    CsrMatStorage *a_, *b_;
    spada_ctx *ctx_ = nullptr;
    CsMat c_;
    spada_stats stats_{};
};

}  // namespace spada
