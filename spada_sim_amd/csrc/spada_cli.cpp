// spada-sim: drop-in command line of the reference (main.rs:30-121, frontend.rs:52-75)
//
//   spada-sim <simulator> <accelerator> <category> <workload> <configuration> [-p|--preprocess]
//             [--preprocess-by length|products] [--output C.mtx|C.bin] [--accumulator lds_hash|sort_merge] [--stats]
//             [--cycle-model]
//
// e.g.  spada-sim accuratesimu spada ss cari config/config_1mb_row1.json
//
// Same positional contract, same JSON configuration keys, same stdout skeleton.  What differs, on
// purpose: the multiply/merge dataflow runs on the GPU instead of being simulated, so no per-task
// `pe: .. cur_cycle: ..` lines exist and the access/cycle counters are measured quantities (see
// spada_host.hpp).  --cycle-model runs the cycle-level Spada model of include/spada_cycle.h on the host instead (no GPU
// involved): the product in the accelerator's order of operations and the simulated counters under the same labels.
// Exit codes: 0 ok, 1 usage error, 101 run-time failure (Rust's panic code).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "spada_cycle.h"
#include "spada_host.hpp"

using namespace spada;

static std::string lower(std::string s)
{
    for (auto &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

// Rust `{:?}` of an f64: shortest digits that round-trip; plain decimal for 1e-5 <= |x| < 1e16.
static std::string rust_f64(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    // buf = d.ddddde[+-]xx
    std::string s(buf);
    size_t epos = s.find('e');
    int exp10 = std::atoi(s.c_str() + epos + 1);
    std::string mant = s.substr(0, epos);
    bool neg = mant[0] == '-';
    if (neg) mant.erase(0, 1);
    std::string digits;
    for (char ch : mant) if (ch != '.') digits += ch;
    std::string out;
    if (exp10 >= 16 || exp10 < -5) {
        out = digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        out += "e" + std::to_string(exp10);
    } else if (exp10 >= 0) {
        if ((int)digits.size() <= exp10 + 1) out = digits + std::string(exp10 + 1 - digits.size(), '0') + ".0";
        else out = digits.substr(0, exp10 + 1) + "." + digits.substr(exp10 + 1);
    } else {
        out = "0." + std::string(-exp10 - 1, '0') + digits;
    }
    return neg ? "-" + out : out;
}

template <class T, class F>
static std::string debug_slice(const std::vector<T> &v, size_t n, F fmt)
{
    std::string s = "[";
    for (size_t i = 0; i < std::min(n, v.size()); ++i) {
        if (i) s += ", ";
        s += fmt(v[i]);
    }
    return s + "]";
}
static std::string u64s(uint64_t x) { return std::to_string(x); }

static void print_gemm(const GEMM &g)   // gemm.rs:56-92, including its habit of printing A's data/indices under --B
{
    const CsMat &a = *g.a, &b = *g.b;
    std::printf("---- %s ----\n", g.name.c_str());
    std::printf("--A: (%llu, %llu)\n", (unsigned long long)a.nrows, (unsigned long long)a.ncols);
    std::printf("data: %s .. \n", debug_slice(a.data, 5, rust_f64).c_str());
    std::printf("indices: %s ...\n", debug_slice(a.indices, 5, u64s).c_str());
    std::printf("indptr: %s ...\n", debug_slice(a.indptr, 5, u64s).c_str());
    std::printf("--B: (%llu, %llu)\n", (unsigned long long)b.nrows, (unsigned long long)b.ncols);
    std::printf("data: %s ...\n", debug_slice(a.data, std::min<size_t>(b.data.size(), 5), rust_f64).c_str());
    std::printf("indices: %s ...\n", debug_slice(a.indices, std::min<size_t>(b.indices.size(), 5), u64s).c_str());
    std::printf("indptr: %s ...\n", debug_slice(b.indptr, 5, u64s).c_str());
    std::printf("\n");
}

static int usage(const char *msg)
{
    std::fprintf(stderr,
                 "error: %s\n\nUSAGE:\n    spada-sim [FLAGS] <simulator> <accelerator> <category> <workload> <configuration>\n\n"
                 "FLAGS:\n    -p, --preprocess    Preprocessing: rows of A sorted by length on the GPU, result mapped back (C is unchanged)\n"
                 "        --preprocess-by <length|products>\n"
                 "        --output <C.mtx|C.bin>    write the product as MatrixMarket, or as a binary CSR dump; prints its checksum\n"
                 "        --checksum          print the checksum line of the product\n"
                 "        --accumulator <lds_hash|sort_merge>\n        --stats             print engine statistics to stderr\n"
                 "        --cycle-model       simulate the accelerator cycle by cycle on the host (include/spada_cycle.h) instead of\n"
                 "                            computing on the GPU: simulated counters, same product\n\n"
                 "ARGS:\n    <simulator>        [possible values: AccurateSimu, TrafficModel, BReuseCounter]\n"
                 "    <accelerator>      [possible values: Ip, Op, MultiRow, Spada]\n"
                 "    <category>         [possible values: SS, NN]\n    <workload>         The workload name\n"
                 "    <configuration>    Configuration file path\n",
                 msg);
    return 1;
}

int main(int argc, char **argv)
{
    std::vector<std::string> pos;
    bool preprocess = false, want_stats = false, want_checksum = false, cycle_model = false;
    std::string output, acc_name, preprocess_by = "length";
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        if (s == "-p" || s == "--preprocess") preprocess = true;
        else if (s == "--stats") want_stats = true;
        else if (s == "--checksum") want_checksum = true;
        else if (s == "--cycle-model") cycle_model = true;
        else if (s == "--preprocess-by" && i + 1 < argc) { preprocess = true; preprocess_by = argv[++i]; }
        else if (s == "--output" && i + 1 < argc) output = argv[++i];
        else if (s == "--accumulator" && i + 1 < argc) acc_name = argv[++i];
        else if (s == "-h" || s == "--help") { usage("help requested"); return 0; }
        else if (s.size() > 1 && s[0] == '-') return usage(("Found argument '" + s + "' which wasn't expected").c_str());
        else pos.push_back(s);
    }
    if (preprocess_by != "length" && preprocess_by != "products") return usage("--preprocess-by must be length or products");
    if (pos.size() != 5) return usage("The following required arguments were not provided (need 5 positionals)");
    const std::string simulator = lower(pos[0]), accel = lower(pos[1]), category = lower(pos[2]);
    const std::string workload = pos[3], configuration = pos[4];
    if (simulator != "accuratesimu" && simulator != "trafficmodel" && simulator != "breusecounter")
        return usage(("'" + pos[0] + "' isn't a valid value for '<simulator>'").c_str());
    Accelerator accelerator;
    if (accel == "ip") accelerator = Accelerator::Ip;
    else if (accel == "op") accelerator = Accelerator::Op;
    else if (accel == "multirow") accelerator = Accelerator::MultiRow;
    else if (accel == "spada") accelerator = Accelerator::Spada;
    else return usage(("'" + pos[1] + "' isn't a valid value for '<accelerator>'").c_str());
    if (category != "ss" && category != "nn") return usage(("'" + pos[2] + "' isn't a valid value for '<category>'").c_str());

    try {
        OmegaConfig cfg = parse_config(configuration);
        int accumulator = cfg.accumulator;
        if (!acc_name.empty()) {
            if (acc_name == "lds_hash") accumulator = SPADA_ACC_LDS_HASH;
            else if (acc_name == "sort_merge") accumulator = SPADA_ACC_SORT_MERGE;
            else return usage("--accumulator must be lds_hash or sort_merge");
        }
        if (category == "nn") {
            // main.rs:35-37 loads a pickled {name: (A, B)} dict through CPython; out of scope here
            std::fprintf(stderr, "error: workload category NN (pickled numpy matrices, py2rust.rs:5-60) is not supported "
                                 "by the GPU front end; convert the matrices to MatrixMarket and use SS\n");
            return 101;
        }
        // py2rust.rs:65,71 banner lines
        std::printf("---- Python Interface ----\n%% Load %s from %s\n", workload.c_str(), cfg.ss_filepath);
        CsMat mat = load_mm_mat(cfg.ss_filepath, workload);
        GEMM gemm = GEMM::from_mat(workload, std::move(mat));

        const uint64_t a_avg = gemm.a->rows() ? gemm.a->nnz() / gemm.a->rows() : 0;
        const uint64_t b_avg = gemm.b->rows() ? gemm.b->nnz() / gemm.b->rows() : 0;
        std::printf("Get GEMM %s\n", gemm.name.c_str());
        print_gemm(gemm);
        std::printf("Avg row len of A: %llu, Avg row len of B: %llu\n", (unsigned long long)a_avg, (unsigned long long)b_avg);

        if (simulator != "accuratesimu") {   // main.rs:119
            std::fprintf(stderr, "Unimplemented simulator %s\n", pos[0].c_str());
            return 101;
        }
        if (cfg.gpus > 1 && !cycle_model) {
            // the binary drives ONE GPU.  The sharded path (A-row blocks, B replicated, RCCL allgatherv of C) is a library interface
            // -- include/spada_comm.h, one process per GPU, see INTEGRATION.md and `bench.py --gpus N` -- so a configuration that asks
            // for more is refused instead of being silently run on one device.  (The cycle-level model below never touches a
            // device: there the key is ignored with a warning.)
            std::fprintf(stderr, "error: configuration key `gpus` = %u: spada-sim runs on one GPU; start one process per GPU and use "
                                 "libspada_comm.so (include/spada_comm.h, INTEGRATION.md) for the sharded path\n", cfg.gpus);
            return 101;
        }
        if (cfg.gpus > 1) std::fprintf(stderr, "warning: configuration key `gpus` = %u is ignored by the cycle-level model (host only)\n", cfg.gpus);
        if (cycle_model) {
            // ---- the cycle-level model (host only) ----------------------------------------------------------------------
            spada_cycle_config cc{};
            cc.struct_size = sizeof(cc);
            cc.pe_num = cfg.pe_num;
            cc.at_num = cfg.at_num;
            cc.lane_num = cfg.lane_num;
            cc.cache_size = cfg.cache_size;
            cc.word_byte = cfg.word_byte;
            cc.block_shape[0] = cfg.block_shape[0];
            cc.block_shape[1] = cfg.block_shape[1];
            cc.mem_latency = cfg.mem_latency;
            cc.cache_latency = cfg.cache_latency;
            cc.freq = cfg.freq;
            cc.channel = cfg.channel;
            cc.bandwidth_per_channel = cfg.bandwidth_per_channel;
            cc.accelerator = accelerator == Accelerator::Ip ? SPADA_ACCEL_IP : accelerator == Accelerator::Op ? SPADA_ACCEL_OP
                             : accelerator == Accelerator::MultiRow ? SPADA_ACCEL_MULTIROW : SPADA_ACCEL_SPADA;
            const spada_csr_view va = gemm.a->view(), vb = gemm.b->view();
            std::vector<uint64_t> remap;
            if (preprocess) {   // preprocessing.rs:76-89: rows by ascending length (or products), stable
                const uint64_t n = va.rows;
                std::vector<uint64_t> key(n, 0);
                for (uint64_t r = 0; r < n; ++r) {
                    if (preprocess_by == "products") {
                        for (uint64_t p = va.indptr[r]; p < va.indptr[r + 1]; ++p)
                            key[r] += vb.indptr[va.indices[p] + 1] - vb.indptr[va.indices[p]];
                    } else {
                        key[r] = va.indptr[r + 1] - va.indptr[r];
                    }
                }
                remap.resize(n);
                for (uint64_t r = 0; r < n; ++r) remap[r] = r;
                std::stable_sort(remap.begin(), remap.end(), [&](uint64_t x, uint64_t y) { return key[x] < key[y]; });
            }
            spada_cycle_model *model = nullptr;
            check(spada_cycle_create(&cc, &va, &vb, preprocess ? remap.data() : nullptr, &model));
            const int rc = spada_cycle_execute(model, 0);
            if (rc != SPADA_OK) {
                spada_cycle_destroy(model);
                check(rc);
            }
            spada_cycle_counts k{};
            k.struct_size = sizeof(k);
            check(spada_cycle_get_counts(model, &k));
            CsMat c;
            c.nrows = va.rows;
            c.ncols = vb.cols;
            c.indptr.assign(va.rows + 1, 0);
            c.indices.assign(k.c_nnz, 0);
            c.data.assign(k.c_nnz, 0.0);
            check(spada_cycle_get_result(model, c.indptr.data(), c.indices.data(), c.data.data()));
            spada_cycle_destroy(model);
            std::printf("-----Result-----\n-----Access count\n");
            std::printf("Execution count: %llu\n", (unsigned long long)k.exec_cycles);
            std::printf("A matrix count: read %llu write %llu\n", (unsigned long long)k.a_read, (unsigned long long)k.a_write);
            std::printf("B matrix count: read %llu write %llu\n", (unsigned long long)k.b_read, (unsigned long long)k.b_write);
            std::printf("C matrix count: read %llu write %llu\n", (unsigned long long)k.c_read, (unsigned long long)k.c_write);
            std::printf("Cache count: read %llu write %llu\n", (unsigned long long)k.cache_read, (unsigned long long)k.cache_write);
            std::printf("-----Output product matrix\n");
            for (uint64_t r = 0; r < std::min<uint64_t>(c.nrows, 10); ++r) {   // storage.rs:115-126
                const std::vector<uint64_t> idx(c.indices.begin() + c.indptr[r], c.indices.begin() + c.indptr[r + 1]);
                const std::vector<double> val(c.data.begin() + c.indptr[r], c.data.begin() + c.indptr[r + 1]);
                std::printf("rowptr: %llu indptr: %s data: %s\n", (unsigned long long)r, debug_slice(idx, 5, u64s).c_str(),
                            debug_slice(val, 5, rust_f64).c_str());
            }
            if (want_stats)
                std::fprintf(stderr, "cycle model: %llu cycles (%llu before the drain discount), cache miss %llu words, evicted B %llu / "
                                     "psum %llu words, %llu blocks, %llu windows, %llu PE merge tasks, %llu tree merge tasks, nnz(C) %llu\n",
                             (unsigned long long)k.exec_cycles, (unsigned long long)k.raw_cycles, (unsigned long long)k.cache_miss,
                             (unsigned long long)k.b_evict, (unsigned long long)k.psum_evict, (unsigned long long)k.blocks,
                             (unsigned long long)k.windows, (unsigned long long)k.pe_merge_tasks,
                             (unsigned long long)k.tree_merge_tasks, (unsigned long long)k.c_nnz);
            if (want_checksum || !output.empty()) {
                spada_csr_view v = c.view();
                spada_checksum cs;
                char line[512];
                check(spada_csr_checksum(&v, &cs));
                check(spada_checksum_format(&cs, line, sizeof line));
                std::printf("-----Checksum of the product matrix\n%s\n", line);
                if (!output.empty()) {
                    const bool bin = output.size() > 4 && output.compare(output.size() - 4, 4, ".bin") == 0;
                    check(bin ? spada_csr_write_bin(output.c_str(), &v) : spada_mtx_write(output.c_str(), &v));
                }
            }
            return 0;
        }
        auto drams = CsrMatStorage::init_with_gemm(gemm);
        if (preprocess) {   // main.rs:60-63
            const int rowmap = preprocess_by == "products" ? sort_by_products(drams.first) : sort_by_length(drams.first);
            drams.first.reorder_row(rowmap);
        }
        const uint64_t output_base_addr = drams.second.indptr().size();
        uint64_t block_shape[2] = {cfg.block_shape[0], cfg.block_shape[1]};
        if (accelerator == Accelerator::Op) { block_shape[0] = cfg.lane_num; block_shape[1] = 1; }   // main.rs:67-72
        Simulator cycle_simu(cfg.pe_num, cfg.at_num, cfg.lane_num, cfg.cache_size, cfg.word_byte, output_base_addr, block_shape,
                             &drams.first, &drams.second, accelerator, cfg.mem_latency, cfg.cache_latency, cfg.freq,
                             cfg.channel, cfg.bandwidth_per_channel, accumulator);
        for (uint32_t r = 0; r < std::max(1u, cfg.repeat); ++r) cycle_simu.execute();

        auto result = cycle_simu.get_exec_result(10);
        auto a_count = cycle_simu.get_a_mat_stat(), b_count = cycle_simu.get_b_mat_stat(), c_count = cycle_simu.get_c_mat_stat();
        auto cache_count = cycle_simu.get_cache_stat();
        std::printf("-----Result-----\n-----Access count\n");
        std::printf("Execution count: %llu\n", (unsigned long long)cycle_simu.get_exec_cycle());
        std::printf("A matrix count: read %llu write %llu\n", (unsigned long long)a_count.first, (unsigned long long)a_count.second);
        std::printf("B matrix count: read %llu write %llu\n", (unsigned long long)b_count.first, (unsigned long long)b_count.second);
        std::printf("C matrix count: read %llu write %llu\n", (unsigned long long)c_count.first, (unsigned long long)c_count.second);
        std::printf("Cache count: read %llu write %llu\n", (unsigned long long)cache_count.first, (unsigned long long)cache_count.second);
        std::printf("-----Output product matrix\n");
        for (const CsrRow &row : result)    // storage.rs:115-126
            std::printf("rowptr: %llu indptr: %s data: %s\n", (unsigned long long)row.rowptr,
                        debug_slice(row.indptr, 5, u64s).c_str(), debug_slice(row.data, 5, rust_f64).c_str());
        if (want_stats) {
            const spada_stats &st = cycle_simu.stats();
            const double ms = st.ms_symbolic_call + st.ms_numeric_call;
            std::fprintf(stderr, "engine: rows %llu nnz(A) %llu products %llu nnz(C) %llu | symbolic %.3f ms numeric %.3f ms | "
                                 "%.3f G nnz(C)/s, algorithmic read %.1f GB/s (%.2f%% of 8 TB/s), rows spilled to HBM scratch %llu\n",
                         (unsigned long long)st.rows, (unsigned long long)st.a_nnz, (unsigned long long)st.nprod,
                         (unsigned long long)st.c_nnz, st.ms_symbolic_call, st.ms_numeric_call, st.c_nnz / ms / 1e6,
                         st.bytes_read / ms / 1e6, st.bytes_read / ms / 1e6 / 8000.0 * 100.0, (unsigned long long)st.spill_rows);
        }
        if (want_checksum || !output.empty()) {
            spada_csr_view v = cycle_simu.result_matrix().view();
            spada_checksum cs;
            char line[512];
            check(spada_csr_checksum(&v, &cs));
            check(spada_checksum_format(&cs, line, sizeof line));
            std::printf("-----Checksum of the product matrix\n%s\n", line);
            if (!output.empty()) {
                const bool bin = output.size() > 4 && output.compare(output.size() - 4, 4, ".bin") == 0;
                check(bin ? spada_csr_write_bin(output.c_str(), &v) : spada_mtx_write(output.c_str(), &v));
            }
        }
    } catch (const Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 101;
    }
    return 0;
}
