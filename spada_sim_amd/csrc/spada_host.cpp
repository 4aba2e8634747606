// Host half of libspada_spgemm.so: error reporting, MatrixMarket ingest, GEMM::from_mat, row-block
// partitioning, JSON configuration and synthetic workloads.  CPU only; nothing here touches HIP.
//
// Reference anchors (into /root/reference/src):
//   load_mm_mat        py2rust.rs:62-97   (scipy.io.mmread(f).tocsr())
//   GEMM::from_mat     gemm.rs:41-53
//   CsrMatStorage      storage.rs:150-160, :214-239
//   parse_config       frontend.rs:8-23, :77-85
//   row blocks         scheduler.rs:296-379 (disjoint A-row blocks)
#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "spada_internal.hpp"

namespace spada {

static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

void clear_error() { g_last_error.clear(); }

// SPADA_TRACE=0/1/2: replaces the reference's trace_println! / trace_print! macros, which a cargo feature compiles in
// or out (util.rs:1-24); here the level is read once from the environment and the lines go to stderr.  1: one line per call
// (what was loaded, what a pipeline run did); 2: the details behind it (row classes, workspace growth, phase times).
int trace_level()
{
    static const int level = [] {
        const char *e = std::getenv("SPADA_TRACE");
        const int v = e ? std::atoi(e) : 0;
        return v < 0 ? 0 : (v > 2 ? 2 : v);
    }();
    return level;
}

void trace(int level, const char *fmt, ...)
{
    if (trace_level() < level) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "[spada trace %d] %s\n", level, buf);
}

// COO triplets -> canonical CSR exactly as coo_matrix.tocsr() leaves it: entries bucketed by row in
// file order, columns sorted inside a row, duplicates summed.
static void coo_to_csr(uint64_t rows, uint64_t cols, const std::vector<uint64_t> &ri,
                       const std::vector<uint64_t> &ci, const std::vector<double> &vv, spada_host_csr &out)
{
    const size_t m = ri.size();
    out.rows = rows;
    out.cols = cols;
    std::vector<uint64_t> start(rows + 1, 0);
    for (size_t e = 0; e < m; ++e) start[ri[e] + 1]++;
    for (uint64_t r = 0; r < rows; ++r) start[r + 1] += start[r];
    std::vector<uint64_t> bc(m);
    std::vector<double> bv(m);
    {
        std::vector<uint64_t> cur(start.begin(), start.end() - 1);
        for (size_t e = 0; e < m; ++e) {
            uint64_t d = cur[ri[e]]++;
            bc[d] = ci[e];
            bv[d] = vv[e];
        }
    }
    out.indptr.assign(rows + 1, 0);
    out.indices.clear();
    out.data.clear();
    out.indices.reserve(m);
    out.data.reserve(m);
    std::vector<uint32_t> perm;
    for (uint64_t r = 0; r < rows; ++r) {
        const uint64_t s = start[r], n = start[r + 1] - s;
        bool sorted = true;
        for (uint64_t j = 1; j < n && sorted; ++j) sorted = bc[s + j - 1] < bc[s + j];
        if (sorted) {
            for (uint64_t j = 0; j < n; ++j) {
                out.indices.push_back(bc[s + j]);
                out.data.push_back(bv[s + j]);
            }
        } else {
            perm.resize(n);
            std::iota(perm.begin(), perm.end(), 0u);
            std::stable_sort(perm.begin(), perm.end(),
                             [&](uint32_t x, uint32_t y) { return bc[s + x] < bc[s + y]; });
            uint64_t j = 0;
            while (j < n) {
                uint64_t c = bc[s + perm[j]];
                double acc = bv[s + perm[j]];
                ++j;
                while (j < n && bc[s + perm[j]] == c) acc += bv[s + perm[j++]];
                out.indices.push_back(c);
                out.data.push_back(acc);
            }
        }
        out.indptr[r + 1] = out.indices.size();
    }
}

// ---- SplitMix64, counter based: value k of stream `seed` is mix(seed + k * golden) ---------------
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed, uint64_t stream = 0) : s(mix64(seed ^ mix64(stream))) {}
    uint64_t next() { s += 0x9E3779B97F4A7C15ull; return mix64(s); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
    uint64_t below(uint64_t n) { return (uint64_t)(uniform() * (double)n); }
    double value() { return 0.1 + 0.9 * uniform(); }                                   // [0.1,1.0)
    uint64_t geometric(double p)  // support 1,2,...
    {
        double u = uniform();
        return 1 + (uint64_t)std::floor(std::log1p(-u) / std::log1p(-p));
    }
    double normal()
    {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

// truncated Zipf over 1..n with exponent s, sampled by inverse CDF
struct ZipfTable {
    std::vector<double> cdf;
    ZipfTable(uint64_t n, double s)
    {
        cdf.resize(n);
        double acc = 0;
        for (uint64_t k = 1; k <= n; ++k) { acc += std::pow((double)k, -s); cdf[k - 1] = acc; }
        for (auto &c : cdf) c /= acc;
    }
    uint64_t sample(double u) const  // returns 1..n
    {
        return (uint64_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) + 1;
    }
};

static void gen_rmat(uint64_t scale, uint64_t ef, uint64_t seed, spada_host_csr &out)
{
    const uint64_t n = 1ull << scale, m = ef * n;
    std::vector<uint64_t> ri(m), ci(m);
    std::vector<double> vv(m);
    const double a = 0.57, b = 0.19, c = 0.19;
#pragma omp parallel for schedule(static)
    for (uint64_t e = 0; e < m; ++e) {
        Rng r(seed, e);
        uint64_t i = 0, j = 0;
        for (uint64_t l = 0; l < scale; ++l) {
            double u = r.uniform();
            uint64_t bi = 0, bj = 0;
            if (u < a) { }
            else if (u < a + b) bj = 1;
            else if (u < a + b + c) bi = 1;
            else { bi = 1; bj = 1; }
            i = (i << 1) | bi;
            j = (j << 1) | bj;
        }
        ri[e] = i;
        ci[e] = j;
        vv[e] = r.value();
    }
    coo_to_csr(n, n, ri, ci, vv, out);
}

static void gen_uniform(uint64_t n, uint64_t k, uint64_t seed, spada_host_csr &out)
{
    std::vector<uint64_t> ri(n * k), ci(n * k);
    std::vector<double> vv(n * k);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) {
        Rng r(seed, i);
        for (uint64_t t = 0; t < k; ++t) {
            ri[i * k + t] = i;
            ci[i * k + t] = r.below(n);
            vv[i * k + t] = r.value();
        }
    }
    coo_to_csr(n, n, ri, ci, vv, out);
}

// Web-crawl surrogate (SURVEY 8d "webbase-like"): power-law out-degree (truncated Zipf s=2.1, max
// 4700), pages grouped into sites whose members link to the site's first pages (shared navigation
// targets), the rest of the links go to hubs drawn Zipf(0.8) from the 100 000 pages with the largest
// noisy out-degree (hubs are both popular and link-rich, which is what makes A*A expensive).
// Web-crawl surrogate (SURVEY 8d "webbase-like"), calibrated to the literature counts of webbase-1M within 5 %
// (nnz 3.11 M, 69.5 M products, nnz(C) 51.1 M: tests/test_host_cpu.py).  Pages are grouped into sites (consecutive
// indices, geometric length, mean 64).  The out-degree of a page is  site richness (Zipf 2.2: few link-rich sites)
// x position factor (the first pages of a site are its index pages) x lognormal page noise,  so link-rich pages
// cluster.  90 % of the links are site-local, towards the site's first pages (shared navigation targets); the rest go
// to hubs (Zipf 0.87 over the 100 000 pages with the largest noisy out-degree).  Two things give A*A the output
// overlap real crawls have (compression 1.36): 30 % of the hub links of a page come, in order, from a list shared by
// its whole site, and a hub link is followed by a short run of links to the pages next to the hub (same site, same
// list), so one row selects several B rows with largely equal link sets.
static void gen_webbase_like(uint64_t n, uint64_t nnz_target, uint64_t seed, spada_host_csr &out)
{
    if (n == 0) n = 1000005;
    if (nnz_target == 0) nnz_target = 3105536;
    const uint64_t dmax = std::min<uint64_t>(4700, n);
    const double p_local = 0.9, p_tmpl = 0.3, p_run = 0.3;
    std::vector<uint64_t> site_start(n), site_end(n);
    std::vector<uint32_t> site_id(n);
    std::vector<double> rich;
    {
        Rng r(seed ^ 0x3333, 0);
        const ZipfTable richz(2000, 2.2);
        uint64_t s0 = 0;
        while (s0 < n) {
            uint64_t len = r.geometric(1.0 / 64.0), s1 = std::min(n, s0 + len);
            for (uint64_t i = s0; i < s1; ++i) { site_start[i] = s0; site_end[i] = s1; site_id[i] = (uint32_t)rich.size(); }
            rich.push_back((double)richz.sample(r.uniform()));
            s0 = s1;
        }
    }
    std::vector<double> raw(n), key(n);
    double total_raw = 0;
    for (uint64_t i = 0; i < n; ++i) {
        Rng r(seed ^ 0x1111, i);
        raw[i] = std::exp(0.5 * r.normal()) * rich[site_id[i]] * (1.0 + 3.0 * std::exp(-(double)(i - site_start[i]) / 3.0));
        key[i] = std::log(raw[i]) + 2.0 * r.normal();
        total_raw += raw[i];
    }
    // the raw edge count exceeds the target by the share duplicate links remove again
    const double scale = 1.77 * (double)nnz_target / total_raw;
    std::vector<uint32_t> deg(n);
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        double d = raw[i] * scale;
        Rng r(seed ^ 0x2222, i);
        uint32_t di = (uint32_t)std::min<double>(d, (double)dmax);
        if (r.uniform() < d - di) ++di;
        deg[i] = std::max<uint32_t>(1, std::min<uint32_t>(di, (uint32_t)dmax));
        total += deg[i];
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return key[x] != key[y] ? key[x] > key[y] : x < y;
    });
    const uint64_t H = std::min<uint64_t>(100000, n);
    ZipfTable hubz(H, 0.87);
    std::vector<uint64_t> ri(total), ci(total), off(n + 1, 0);
    std::vector<double> vv(total);
    for (uint64_t i = 0; i < n; ++i) off[i + 1] = off[i] + deg[i];
#pragma omp parallel for schedule(dynamic, 1024)
    for (uint64_t i = 0; i < n; ++i) {
        Rng r(seed ^ 0x4444, i);
        uint64_t tmpl_used = 0, run_next = 0;
        bool in_run = false;
        for (uint64_t t = 0; t < deg[i]; ++t) {
            uint64_t e = off[i] + t, j;
            if (in_run && run_next < n && r.uniform() < p_run) {
                j = run_next++;                                   // next page of the hub's site
            } else if (r.uniform() < p_local) {
                j = std::min(site_start[i] + r.geometric(1.0 / 6.0) - 1, site_end[i] - 1);
                in_run = false;
            } else {
                if (r.uniform() < p_tmpl) {                       // the site's shared link list, in order
                    Rng rt(seed ^ 0x7777, ((uint64_t)site_id[i] << 20) + tmpl_used++);
                    j = order[hubz.sample(rt.uniform()) - 1];
                } else {
                    j = order[hubz.sample(r.uniform()) - 1];
                }
                in_run = true;
                run_next = j + 1;
            }
            ri[e] = i;
            ci[e] = j;
            vv[e] = r.value();
        }
    }
    coo_to_csr(n, n, ri, ci, vv, out);
}

// FEM-like symmetric surrogate (cop20k_A): nodes of a 3-D grid, every node has a weight (lognormal: refined and coarse
// regions), an undirected edge to a node of the 26-neighbourhood exists with probability p1 * w_i * w_j, to a node of the
// second ring (|d| <= 2) with p2 * w_i * w_j.  Neighbours of neighbours overlap heavily, which is what gives the real
// matrix its compression of 4.3 in A*A.
static void gen_cop20k_like(uint64_t n, uint64_t seed, spada_host_csr &out)
{
    if (n == 0) n = 121192;
    const uint64_t nx = 46, ny = 52;   // x fastest; 46 * 52 * 51 >= 121192
    const double p1 = 0.62, p2 = 0.0225, wsig = 0.85;
    std::vector<double> w(n);
    for (uint64_t i = 0; i < n; ++i) {
        // weights vary smoothly in space: one draw per 4 x 4 x 4 cell
        const uint64_t x = i % nx, y = (i / nx) % ny, z = i / (nx * ny);
        Rng r(seed ^ 0x6666, ((z / 4) << 40) | ((y / 4) << 20) | (x / 4));
        w[i] = std::exp(wsig * r.normal());
    }
    std::vector<uint64_t> ri, ci;
    std::vector<double> vv;
    ri.reserve(n * 24);
    ci.reserve(n * 24);
    vv.reserve(n * 24);
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t x = (int64_t)(i % nx), y = (int64_t)((i / nx) % ny), z = (int64_t)(i / (nx * ny));
        {
            Rng r(seed, i);
            ri.push_back(i); ci.push_back(i); vv.push_back(r.value());
        }
        for (int64_t dz = -2; dz <= 2; ++dz)
            for (int64_t dy = -2; dy <= 2; ++dy)
                for (int64_t dx = -2; dx <= 2; ++dx) {
                    if (!dx && !dy && !dz) continue;
                    const int64_t xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || yy < 0 || zz < 0 || xx >= (int64_t)nx || yy >= (int64_t)ny) continue;
                    const uint64_t j = (uint64_t)zz * nx * ny + (uint64_t)yy * nx + (uint64_t)xx;
                    if (j >= n || j < i) continue;   // every undirected edge is decided once, by its lower end
                    const bool ring1 = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                    Rng r(seed ^ 0x8888, i * 125 + (uint64_t)((dz + 2) * 25 + (dy + 2) * 5 + (dx + 2)));
                    if (r.uniform() >= (ring1 ? p1 : p2) * w[i] * w[j]) continue;
                    const double v = r.value();
                    ri.push_back(i); ci.push_back(j); vv.push_back(v);
                    ri.push_back(j); ci.push_back(i); vv.push_back(v);
                }
    }
    coo_to_csr(n, n, ri, ci, vv, out);
}

// cage12-like: the column offsets of a row are drawn (without replacement, each with probability q_i) from a fixed
// two-generator lattice { x * d1 + y * d2 : |x| <= 3, |y| <= 2 } inside a +-n/50 locality radius -- sums of two
// offsets fall on the (13 x 9)-point lattice again, so A*A has ~117 entries per row from ~260 products, like the
// real matrix (compression 2.3).
static void gen_cage12_like(uint64_t n, uint64_t seed, spada_host_csr &out)
{
    if (n == 0) n = 130228;
    const int64_t d1 = std::max<int64_t>(1, (int64_t)(n / 3500)), d2 = std::max<int64_t>(2, (int64_t)(n / 160));
    const double q = 0.226, qsig = 0.4;
    std::vector<uint64_t> ri, ci;
    std::vector<double> vv;
    ri.reserve(n * 18);
    ci.reserve(n * 18);
    vv.reserve(n * 18);
    for (uint64_t i = 0; i < n; ++i) {
        Rng r(seed, i);
        ri.push_back(i); ci.push_back(i); vv.push_back(r.value());
        const double qi = std::min(1.0, q * std::exp(qsig * r.normal()));
        for (int64_t yy = -(int64_t)3; yy <= (int64_t)3; ++yy)
            for (int64_t xx = -(int64_t)4; xx <= (int64_t)4; ++xx) {
                if (!xx && !yy) continue;
                const double u = r.uniform(), v = r.value();
                const int64_t j = (int64_t)i + xx * d1 + yy * d2;
                if (u >= qi || j < 0 || j >= (int64_t)n) continue;
                ri.push_back(i); ci.push_back((uint64_t)j); vv.push_back(v);
            }
    }
    coo_to_csr(n, n, ri, ci, vv, out);
}

// mc2depi-like: 2-D grid Markov chain, <= 4 transitions per state: (x-1,y), (x+1,y), (x-1,y+1), (x,y+1).
static void gen_mc2depi_like(uint64_t n, uint64_t seed, spada_host_csr &out)
{
    if (n == 0) n = 525825;   // = 779 * 675
    uint64_t W = 779;
    if (n % W != 0) W = (uint64_t)std::max(1.0, std::floor(std::sqrt((double)n)));
    std::vector<uint64_t> ri, ci;
    std::vector<double> vv;
    ri.reserve(4 * n);
    ci.reserve(4 * n);
    vv.reserve(4 * n);
    for (uint64_t i = 0; i < n; ++i) {
        Rng r(seed, i);
        uint64_t x = i % W;
        auto add = [&](int64_t j) {
            if (j >= 0 && j < (int64_t)n) { ri.push_back(i); ci.push_back((uint64_t)j); vv.push_back(r.value()); }
        };
        if (x > 0) add((int64_t)i - 1);
        if (x + 1 < W) add((int64_t)i + 1);
        if (x > 0) add((int64_t)(i + W) - 1);
        add((int64_t)(i + W));
    }
    coo_to_csr(n, n, ri, ci, vv, out);
}

// ---- MatrixMarket ---------------------------------------------------------------------------------
static std::string lower(std::string s)
{
    for (auto &ch : s) ch = (char)std::tolower((unsigned char)ch);
    return s;
}

static int mtx_read(const char *path, spada_host_csr &out)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(SPADA_ERR_IO, "cannot open %s: %s", path, std::strerror(errno));
    std::string text;
    {
        char buf[1 << 16];
        size_t k;
        while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
        std::fclose(f);
    }
    const char *p = text.c_str(), *end = p + text.size();
    auto next_line = [&](std::string &line) -> bool {
        if (p >= end) return false;
        const char *q = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        if (!q) q = end;
        line.assign(p, q);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        p = (q < end) ? q + 1 : end;
        return true;
    };
    std::string line;
    if (!next_line(line)) return fail(SPADA_ERR_PARSE, "%s: empty file", path);
    std::istringstream hs(line);
    std::string banner, object, format, field, symmetry;
    hs >> banner >> object >> format >> field >> symmetry;
    if (lower(banner) != "%%matrixmarket" || lower(object) != "matrix")
        return fail(SPADA_ERR_PARSE, "%s: not a MatrixMarket matrix header: '%s'", path, line.c_str());
    format = lower(format);
    field = lower(field);
    symmetry = lower(symmetry);
    if (format != "coordinate")
        return fail(SPADA_ERR_UNSUPPORTED,
                    "%s: format '%s' is not sparse (the reference calls .tocsr() on mmread's result, "
                    "py2rust.rs:74, which only exists for coordinate files)", path, format.c_str());
    const bool pattern = field == "pattern";
    if (!(pattern || field == "real" || field == "double" || field == "integer" || field == "unsigned-integer"))
        return fail(SPADA_ERR_UNSUPPORTED, "%s: field '%s' cannot be stored as f64", path, field.c_str());
    int sym;  // 0 general, 1 symmetric, 2 skew
    if (symmetry == "general") sym = 0;
    else if (symmetry == "symmetric" || symmetry == "hermitian") sym = 1;
    else if (symmetry == "skew-symmetric") sym = 2;
    else return fail(SPADA_ERR_PARSE, "%s: unknown symmetry '%s'", path, symmetry.c_str());

    // size line: first line that is neither a comment nor blank
    uint64_t rows = 0, cols = 0, nent = 0;
    for (;;) {
        if (!next_line(line)) return fail(SPADA_ERR_PARSE, "%s: missing size line", path);
        size_t k = line.find_first_not_of(" \t");
        if (k == std::string::npos || line[k] == '%') continue;
        unsigned long long r_, c_, n_;
        if (std::sscanf(line.c_str(), "%llu %llu %llu", &r_, &c_, &n_) != 3)
            return fail(SPADA_ERR_PARSE, "%s: bad size line '%s'", path, line.c_str());
        rows = r_; cols = c_; nent = n_;
        break;
    }
    std::vector<uint64_t> ri, ci;
    std::vector<double> vv;
    // (an entry takes at least four bytes of text: a size line that announces more than the file can hold is refused before
    // anything is reserved for it)
    if (nent > (uint64_t)text.size() / 4 + 1)
        return fail(SPADA_ERR_PARSE, "%s: size line announces %llu entries, the file has %llu bytes", path, (unsigned long long)nent,
                    (unsigned long long)text.size());
    const size_t cap = (size_t)nent * (sym ? 2 : 1);
    ri.reserve(cap);
    ci.reserve(cap);
    vv.reserve(cap);
    uint64_t got = 0;
    while (got < nent) {
        // skip whitespace / blank lines / comments between entries
        while (p < end && std::isspace((unsigned char)*p)) ++p;
        if (p >= end) break;
        if (*p == '%') { while (p < end && *p != '\n') ++p; continue; }
        char *q;
        errno = 0;
        unsigned long long i = std::strtoull(p, &q, 10);
        if (q == p) return fail(SPADA_ERR_PARSE, "%s: bad row index at entry %llu", path, (unsigned long long)got + 1);
        p = q;
        unsigned long long j = std::strtoull(p, &q, 10);
        if (q == p) return fail(SPADA_ERR_PARSE, "%s: bad column index at entry %llu", path, (unsigned long long)got + 1);
        p = q;
        double v = 1.0;
        if (!pattern) {
            v = std::strtod(p, &q);
            if (q == p) return fail(SPADA_ERR_PARSE, "%s: bad value at entry %llu", path, (unsigned long long)got + 1);
            p = q;
        }
        if (i < 1 || i > rows || j < 1 || j > cols)
            return fail(SPADA_ERR_PARSE, "%s: entry %llu (%llu,%llu) outside %llux%llu", path,
                        (unsigned long long)got + 1, i, j, (unsigned long long)rows, (unsigned long long)cols);
        ri.push_back(i - 1);
        ci.push_back(j - 1);
        vv.push_back(v);
        if (sym && i != j) {
            ri.push_back(j - 1);
            ci.push_back(i - 1);
            vv.push_back(sym == 2 ? -v : v);
        }
        ++got;
    }
    if (got != nent)
        return fail(SPADA_ERR_PARSE, "%s: %llu entries announced, %llu found", path, (unsigned long long)nent,
                    (unsigned long long)got);
    if (sym && rows != cols) return fail(SPADA_ERR_PARSE, "%s: symmetric matrix must be square", path);
    coo_to_csr(rows, cols, ri, ci, vv, out);
    return SPADA_OK;
}

static int validate(const spada_csr_view *m, const char *what)
{
    if (!m) return fail(SPADA_ERR_INVALID, "%s: null view", what);
    if (!m->indptr) return fail(SPADA_ERR_INVALID, "%s: null indptr", what);
    if (m->nnz && (!m->indices || !m->data)) return fail(SPADA_ERR_INVALID, "%s: null indices/data", what);
    if (m->indptr[0] != 0) return fail(SPADA_ERR_INVALID, "%s: indptr[0] != 0", what);
    if (m->indptr[m->rows] != m->nnz)
        return fail(SPADA_ERR_INVALID, "%s: indptr[rows] = %llu but nnz = %llu", what,
                    (unsigned long long)m->indptr[m->rows], (unsigned long long)m->nnz);
    for (uint64_t r = 0; r < m->rows; ++r) {
        uint64_t s = m->indptr[r], e = m->indptr[r + 1];
        if (e < s || e > m->nnz) return fail(SPADA_ERR_INVALID, "%s: indptr not monotone at row %llu", what, (unsigned long long)r);
        for (uint64_t q = s; q < e; ++q) {
            if (m->indices[q] >= m->cols)
                return fail(SPADA_ERR_INVALID, "%s: column %llu >= cols %llu in row %llu", what,
                            (unsigned long long)m->indices[q], (unsigned long long)m->cols, (unsigned long long)r);
            if (q > s && m->indices[q - 1] >= m->indices[q])
                return fail(SPADA_ERR_INVALID, "%s: columns of row %llu not ascending/unique", what, (unsigned long long)r);
        }
    }
    return SPADA_OK;
}

static void transpose(const spada_csr_view &a, spada_host_csr &t)
{
    t.rows = a.cols;
    t.cols = a.rows;
    t.indptr.assign(a.cols + 1, 0);
    t.indices.resize(a.nnz);
    t.data.resize(a.nnz);
    for (uint64_t q = 0; q < a.nnz; ++q) t.indptr[a.indices[q] + 1]++;
    for (uint64_t c = 0; c < a.cols; ++c) t.indptr[c + 1] += t.indptr[c];
    std::vector<uint64_t> cur(t.indptr.begin(), t.indptr.end() - 1);
    for (uint64_t r = 0; r < a.rows; ++r)
        for (uint64_t q = a.indptr[r]; q < a.indptr[r + 1]; ++q) {
            uint64_t d = cur[a.indices[q]]++;
            t.indices[d] = r;
            t.data[d] = a.data[q];
        }
}

// ---- minimal JSON (objects, arrays, strings, numbers, true/false/null) ------------------------------
struct JVal {
    enum { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    double num = 0;
    bool integral = false;
    std::string str;
    std::vector<JVal> arr;
    std::map<std::string, JVal> obj;
};
struct JParser {
    const char *p, *e;
    std::string err;
    void ws() { while (p < e && std::isspace((unsigned char)*p)) ++p; }
    bool parse(JVal &v)
    {
        ws();
        if (p >= e) { err = "unexpected end"; return false; }
        if (*p == '{') {
            v.t = JVal::OBJ; ++p; ws();
            if (p < e && *p == '}') { ++p; return true; }
            for (;;) {
                JVal k;
                ws();
                if (p >= e || *p != '"' || !parse(k)) { if (err.empty()) err = "expected key string"; return false; }
                ws();
                if (p >= e || *p != ':') { err = "expected ':'"; return false; }
                ++p;
                JVal x;
                if (!parse(x)) return false;
                v.obj[k.str] = std::move(x);
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; return true; }
                err = "expected ',' or '}'";
                return false;
            }
        }
        if (*p == '[') {
            v.t = JVal::ARR; ++p; ws();
            if (p < e && *p == ']') { ++p; return true; }
            for (;;) {
                JVal x;
                if (!parse(x)) return false;
                v.arr.push_back(std::move(x));
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; return true; }
                err = "expected ',' or ']'";
                return false;
            }
        }
        if (*p == '"') {
            v.t = JVal::STR; ++p;
            while (p < e && *p != '"') {
                if (*p == '\\' && p + 1 < e) {
                    ++p;
                    switch (*p) {
                        case 'n': v.str += '\n'; break;
                        case 't': v.str += '\t'; break;
                        case 'r': v.str += '\r'; break;
                        case 'b': v.str += '\b'; break;
                        case 'f': v.str += '\f'; break;
                        case 'u': {
                            if (p + 4 >= e) { err = "bad \\u escape"; return false; }
                            unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                            if (cp < 0x80) v.str += (char)cp;
                            else if (cp < 0x800) { v.str += (char)(0xC0 | (cp >> 6)); v.str += (char)(0x80 | (cp & 0x3F)); }
                            else { v.str += (char)(0xE0 | (cp >> 12)); v.str += (char)(0x80 | ((cp >> 6) & 0x3F)); v.str += (char)(0x80 | (cp & 0x3F)); }
                            p += 4;
                            break;
                        }
                        default: v.str += *p;
                    }
                    ++p;
                } else v.str += *p++;
            }
            if (p >= e) { err = "unterminated string"; return false; }
            ++p;
            return true;
        }
        if (!std::strncmp(p, "true", 4)) { v.t = JVal::BOOL; v.num = 1; p += 4; return true; }
        if (!std::strncmp(p, "false", 5)) { v.t = JVal::BOOL; v.num = 0; p += 5; return true; }
        if (!std::strncmp(p, "null", 4)) { v.t = JVal::NUL; p += 4; return true; }
        char *q;
        v.num = std::strtod(p, &q);
        if (q == p) { err = std::string("unexpected character '") + *p + "'"; return false; }
        v.t = JVal::NUM;
        v.integral = true;
        for (const char *c = p; c < q; ++c) if (*c == '.' || *c == 'e' || *c == 'E') v.integral = false;
        p = q;
        return true;
    }
};

}  // namespace spada

using namespace spada;

// No C++ exception crosses the C ABI (the reference's panic! => process exit is not copied): sizes that come from
// files or callers may make an allocation throw; the entry points that allocate run their body through this.
template <class F>
static int guarded(const char *what, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(SPADA_ERR_OOM, "%s: out of host memory", what);
    } catch (const std::length_error &) {
        return fail(SPADA_ERR_OOM, "%s: size exceeds what the host can allocate", what);
    } catch (const std::exception &e) {
        return fail(SPADA_ERR_INVALID, "%s: %s", what, e.what());
    }
}

extern "C" {

const char *spada_last_error(void) { return g_last_error.c_str(); }
int spada_abi_version(void) { return SPADA_ABI_VERSION; }

int spada_mtx_read(const char *path, spada_host_csr **out)
{
    return guarded("spada_mtx_read", [&]() -> int {
        if (!path || !out) return fail(SPADA_ERR_INVALID, "spada_mtx_read: null argument");
        *out = nullptr;
        auto m = std::make_unique<spada_host_csr>();
        int rc = mtx_read(path, *m);
        if (rc) return rc;
        trace(1, "load_mm_mat %s: %llu x %llu, %llu entries", path, (unsigned long long)m->rows, (unsigned long long)m->cols,
              (unsigned long long)m->nnz());
        *out = m.release();
        return SPADA_OK;
    });
}

// FNV-1a, 64 bit, over a byte range
static uint64_t fnv1a(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 0x100000001B3ull;
    }
    return h;
}

int spada_csr_checksum(const spada_csr_view *m, spada_checksum *out)
{
    if (!out) return fail(SPADA_ERR_INVALID, "spada_csr_checksum: null out");
    int rc = validate(m, "spada_csr_checksum");
    if (rc) return rc;
    out->rows = m->rows;
    out->cols = m->cols;
    out->nnz = m->nnz;
    uint64_t h = 0xCBF29CE484222325ull;
    h = fnv1a(h, m->indptr, (m->rows + 1) * 8);
    h = fnv1a(h, m->indices, m->nnz * 8);
    out->structure_hash = h;
    out->value_hash = fnv1a(0xCBF29CE484222325ull, m->data, m->nnz * 8);
    double s = 0.0, sa = 0.0;
    for (uint64_t q = 0; q < m->nnz; ++q) {   // in storage order, one rounding per addition
        s += m->data[q];
        sa += std::fabs(m->data[q]);
    }
    out->value_sum = s;
    out->value_abs_sum = sa;
    return SPADA_OK;
}

int spada_checksum_format(const spada_checksum *cs, char *buf, size_t n)
{
    if (!cs || !buf || !n) return fail(SPADA_ERR_INVALID, "spada_checksum_format: null argument");
    std::snprintf(buf, n, "rows %llu cols %llu nnz %llu structure %016llx values %016llx sum %.17g abs_sum %.17g",
                  (unsigned long long)cs->rows, (unsigned long long)cs->cols, (unsigned long long)cs->nnz,
                  (unsigned long long)cs->structure_hash, (unsigned long long)cs->value_hash, cs->value_sum, cs->value_abs_sum);
    return SPADA_OK;
}

// binary CSR dump: "SPADACSR" | version u64 | rows | cols | nnz | indptr u64[rows + 1] | indices u64[nnz] | data f64[nnz] |
// structure_hash | value_hash   (little endian, the in-memory layout of CsrMatStorage's three Vecs, storage.rs:150-160)
static const char BIN_MAGIC[8] = {'S', 'P', 'A', 'D', 'A', 'C', 'S', 'R'};

int spada_csr_write_bin(const char *path, const spada_csr_view *m)
{
    return guarded("spada_csr_write_bin", [&]() -> int {
        if (!path) return fail(SPADA_ERR_INVALID, "spada_csr_write_bin: null path");
        spada_checksum cs;
        int rc = spada_csr_checksum(m, &cs);
        if (rc) return rc;
        FILE *f = std::fopen(path, "wb");
        if (!f) return fail(SPADA_ERR_IO, "cannot create %s: %s", path, std::strerror(errno));
        const uint64_t hdr[4] = {1, m->rows, m->cols, m->nnz}, tail[2] = {cs.structure_hash, cs.value_hash};
        bool ok = std::fwrite(BIN_MAGIC, 1, 8, f) == 8 && std::fwrite(hdr, 8, 4, f) == 4 &&
                  std::fwrite(m->indptr, 8, m->rows + 1, f) == m->rows + 1 && std::fwrite(m->indices, 8, m->nnz, f) == m->nnz &&
                  std::fwrite(m->data, 8, m->nnz, f) == m->nnz && std::fwrite(tail, 8, 2, f) == 2;
        if (std::fclose(f) != 0) ok = false;
        if (!ok) return fail(SPADA_ERR_IO, "write to %s failed", path);
        return SPADA_OK;
    });
}

int spada_csr_read_bin(const char *path, spada_host_csr **out)
{
    return guarded("spada_csr_read_bin", [&]() -> int {
        if (!path || !out) return fail(SPADA_ERR_INVALID, "spada_csr_read_bin: null argument");
        *out = nullptr;
        FILE *f = std::fopen(path, "rb");
        if (!f) return fail(SPADA_ERR_IO, "cannot open %s: %s", path, std::strerror(errno));
        char magic[8];
        uint64_t hdr[4], tail[2];
        auto m = std::make_unique<spada_host_csr>();
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, BIN_MAGIC, 8) == 0 && std::fread(hdr, 8, 4, f) == 4 && hdr[0] == 1;
        if (ok) {
            m->rows = hdr[1];
            m->cols = hdr[2];
            // the header is not trusted: the file must have exactly the size it announces before anything is allocated from it
        struct stat fst;
        if (hdr[1] > (1ull << 40) || hdr[3] > (1ull << 44) || fstat(fileno(f), &fst) != 0 ||
            (unsigned long long)fst.st_size != 8ull + 32ull + (hdr[1] + 1) * 8ull + hdr[3] * 16ull + 16ull)
            ok = false;
        }
        if (ok) {
            m->indptr.resize(hdr[1] + 1);
            m->indices.resize(hdr[3]);
            m->data.resize(hdr[3]);
            ok = std::fread(m->indptr.data(), 8, hdr[1] + 1, f) == hdr[1] + 1 && std::fread(m->indices.data(), 8, hdr[3], f) == hdr[3] &&
                 std::fread(m->data.data(), 8, hdr[3], f) == hdr[3] && std::fread(tail, 8, 2, f) == 2;
        }
        std::fclose(f);
        if (!ok) return fail(SPADA_ERR_PARSE, "%s is not a complete SPADACSR version 1 file", path);
        const spada_csr_view v = view_of(*m);
        spada_checksum cs;
        if (spada_csr_checksum(&v, &cs)) {
            const std::string why = spada_last_error();
            return fail(SPADA_ERR_PARSE, "%s: corrupt CSR (%s)", path, why.c_str());
        }
        if (cs.structure_hash != tail[0] || cs.value_hash != tail[1]) return fail(SPADA_ERR_PARSE, "%s: checksum mismatch", path);
        *out = m.release();
        return SPADA_OK;
    });
}

int spada_mtx_write(const char *path, const spada_csr_view *m)
{
    return guarded("spada_mtx_write", [&]() -> int {
        if (!path) return fail(SPADA_ERR_INVALID, "spada_mtx_write: null path");
        int rc = validate(m, "spada_mtx_write");
        if (rc) return rc;
        spada_checksum cs;
        if ((rc = spada_csr_checksum(m, &cs))) return rc;
        char line[512];
        (void)spada_checksum_format(&cs, line, sizeof line);
        FILE *f = std::fopen(path, "wb");
        if (!f) return fail(SPADA_ERR_IO, "cannot create %s: %s", path, std::strerror(errno));
        // the checksum travels as a MatrixMarket comment: any reader skips it, `spada-sim --checksum` prints the same line
        std::fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%% spada-sim checksum: %s\n%llu %llu %llu\n", line,
                     (unsigned long long)m->rows, (unsigned long long)m->cols, (unsigned long long)m->nnz);
        for (uint64_t r = 0; r < m->rows; ++r)
            for (uint64_t q = m->indptr[r]; q < m->indptr[r + 1]; ++q)
                std::fprintf(f, "%llu %llu %.17g\n", (unsigned long long)r + 1, (unsigned long long)m->indices[q] + 1, m->data[q]);
        if (std::fclose(f) != 0) return fail(SPADA_ERR_IO, "write to %s failed", path);
        return SPADA_OK;
    });
}

int spada_host_csr_from_view(const spada_csr_view *v, spada_host_csr **out)
{
    return guarded("spada_host_csr_from_view", [&]() -> int {
        if (!out) return fail(SPADA_ERR_INVALID, "spada_host_csr_from_view: null out");
        *out = nullptr;
        int rc = validate(v, "spada_host_csr_from_view");
        if (rc) return rc;
        auto m = std::make_unique<spada_host_csr>();
        m->rows = v->rows;
        m->cols = v->cols;
        m->indptr.assign(v->indptr, v->indptr + v->rows + 1);
        m->indices.assign(v->indices, v->indices + v->nnz);
        m->data.assign(v->data, v->data + v->nnz);
        *out = m.release();
        return SPADA_OK;
    });
}

int spada_host_csr_view(const spada_host_csr *m, spada_csr_view *out)
{
    if (!m || !out) return fail(SPADA_ERR_INVALID, "spada_host_csr_view: null argument");
    *out = view_of(*m);
    return SPADA_OK;
}

void spada_host_csr_free(spada_host_csr *m) { delete m; }

int spada_csr_validate(const spada_csr_view *m) { return validate(m, "csr"); }

int spada_transpose(const spada_csr_view *a, spada_host_csr **out)
{
    return guarded("spada_transpose", [&]() -> int {
        if (!out) return fail(SPADA_ERR_INVALID, "spada_transpose: null out");
        *out = nullptr;
        int rc = validate(a, "spada_transpose");
        if (rc) return rc;
        auto t = std::make_unique<spada_host_csr>();
        transpose(*a, *t);
        *out = t.release();
        return SPADA_OK;
    });
}

int spada_from_mat(const spada_csr_view *a, spada_host_csr **b_out, int *b_is_a)
{
    if (!a || !b_out || !b_is_a) return fail(SPADA_ERR_INVALID, "spada_from_mat: null argument");
    *b_out = nullptr;
    if (a->rows == a->cols) {   // gemm.rs:43-44
        int rc = validate(a, "spada_from_mat");
        if (rc) return rc;
        *b_is_a = 1;
        return SPADA_OK;
    }
    *b_is_a = 0;                // gemm.rs:45-47
    return spada_transpose(a, b_out);
}

int spada_count_products(const spada_csr_view *a, const spada_csr_view *b, uint64_t r0, uint64_t r1, uint64_t *nprod)
{
    if (!a || !b || !nprod) return fail(SPADA_ERR_INVALID, "spada_count_products: null argument");
    if (a->cols != b->rows) return fail(SPADA_ERR_INVALID, "inner dimensions differ: A.cols=%llu B.rows=%llu",
                                        (unsigned long long)a->cols, (unsigned long long)b->rows);
    if (r0 > r1 || r1 > a->rows) return fail(SPADA_ERR_INVALID, "bad row range");
    uint64_t n = 0;
    for (uint64_t q = a->indptr[r0]; q < a->indptr[r1]; ++q) {
        uint64_t k = a->indices[q];
        if (k >= b->rows) return fail(SPADA_ERR_INVALID, "A column %llu >= B.rows", (unsigned long long)k);
        n += b->indptr[k + 1] - b->indptr[k];
    }
    *nprod = n;
    return SPADA_OK;
}

int spada_partition_rows(const spada_csr_view *a, const spada_csr_view *b, uint32_t nparts, uint64_t *bounds)
{
    return guarded("spada_partition_rows", [&]() -> int {
        if (!a || !b || !bounds || nparts == 0) return fail(SPADA_ERR_INVALID, "spada_partition_rows: bad argument");
        if (a->cols != b->rows) return fail(SPADA_ERR_INVALID, "inner dimensions differ");
        // per-row cost = products + row length + 1 (so that empty rows still spread out)
        std::vector<uint64_t> pre(a->rows + 1, 0);
        for (uint64_t r = 0; r < a->rows; ++r) {
            uint64_t w = 1 + (a->indptr[r + 1] - a->indptr[r]);
            for (uint64_t q = a->indptr[r]; q < a->indptr[r + 1]; ++q) {
                uint64_t k = a->indices[q];
                if (k >= b->rows) return fail(SPADA_ERR_INVALID, "A column %llu >= B.rows", (unsigned long long)k);
                w += b->indptr[k + 1] - b->indptr[k];
            }
            pre[r + 1] = pre[r] + w;
        }
        const uint64_t total = pre[a->rows];
        bounds[0] = 0;
        for (uint32_t p = 1; p < nparts; ++p) {
            // smallest r with pre[r] >= total * p / nparts
            long double target = (long double)total * p / nparts;
            uint64_t r = (uint64_t)(std::lower_bound(pre.begin(), pre.end(), (uint64_t)std::ceil((double)target)) - pre.begin());
            r = std::min<uint64_t>(r, a->rows);
            bounds[p] = std::max(r, bounds[p - 1]);
        }
        bounds[nparts] = a->rows;
        return SPADA_OK;
    });
}

int spada_generate(int kind, uint64_t p0, uint64_t p1, uint64_t seed, spada_host_csr **out)
{
    return guarded("spada_generate", [&]() -> int {
        if (!out) return fail(SPADA_ERR_INVALID, "spada_generate: null out");
        *out = nullptr;
        auto m = std::make_unique<spada_host_csr>();
        switch (kind) {
            case SPADA_GEN_RMAT:
                if (p0 < 1 || p0 > 30) return fail(SPADA_ERR_INVALID, "rmat scale %llu out of range", (unsigned long long)p0);
                gen_rmat(p0, p1 ? p1 : 16, seed, *m);
                break;
            case SPADA_GEN_WEBBASE_LIKE: gen_webbase_like(p0, p1, seed, *m); break;
            case SPADA_GEN_COP20K_LIKE: gen_cop20k_like(p0, seed, *m); break;
            case SPADA_GEN_CAGE12_LIKE: gen_cage12_like(p0, seed, *m); break;
            case SPADA_GEN_MC2DEPI_LIKE: gen_mc2depi_like(p0, seed, *m); break;
            case SPADA_GEN_UNIFORM:
                if (p0 == 0) return fail(SPADA_ERR_INVALID, "uniform generator needs rows > 0");
                gen_uniform(p0, p1 ? p1 : 8, seed, *m);
                break;
            default: return fail(SPADA_ERR_INVALID, "unknown generator kind %d", kind);
        }
        *out = m.release();
        return SPADA_OK;
    });
}

int spada_config_parse(const char *path, spada_config *out)
{
    return guarded("spada_config_parse", [&]() -> int {
        if (!path || !out) return fail(SPADA_ERR_INVALID, "spada_config_parse: null argument");
        std::ifstream in(path, std::ios::binary);
        if (!in) return fail(SPADA_ERR_IO, "cannot open %s: %s", path, std::strerror(errno));
        std::stringstream ss;
        ss << in.rdbuf();
        std::string text = ss.str();
        JParser jp{text.c_str(), text.c_str() + text.size(), {}};
        JVal root;
        if (!jp.parse(root)) return fail(SPADA_ERR_PARSE, "%s: JSON error: %s", path, jp.err.c_str());
        jp.ws();
        if (jp.p != jp.e) return fail(SPADA_ERR_PARSE, "%s: trailing characters after JSON value", path);
        if (root.t != JVal::OBJ) return fail(SPADA_ERR_PARSE, "%s: top-level JSON value must be an object", path);
        std::memset(out, 0, sizeof *out);
        auto need = [&](const char *k, int type) -> const JVal * {
            auto it = root.obj.find(k);
            if (it == root.obj.end()) { fail(SPADA_ERR_PARSE, "%s: missing field `%s`", path, k); return nullptr; }
            if ((int)it->second.t != type) { fail(SPADA_ERR_PARSE, "%s: field `%s` has the wrong type", path, k); return nullptr; }
            return &it->second;
        };
        auto get_usize = [&](const char *k, uint64_t &dst) -> bool {
            const JVal *v = need(k, JVal::NUM);
            if (!v) return false;
            if (!v->integral || v->num < 0) { fail(SPADA_ERR_PARSE, "%s: field `%s` must be a non-negative integer", path, k); return false; }
            dst = (uint64_t)v->num;
            return true;
        };
        auto get_f32 = [&](const char *k, float &dst) -> bool {
            const JVal *v = need(k, JVal::NUM);
            if (!v) return false;
            dst = (float)v->num;
            return true;
        };
        auto get_str = [&](const char *k, char *dst, size_t cap) -> bool {
            const JVal *v = need(k, JVal::STR);
            if (!v) return false;
            if (v->str.size() >= cap) { fail(SPADA_ERR_PARSE, "%s: field `%s` too long", path, k); return false; }
            std::memcpy(dst, v->str.c_str(), v->str.size() + 1);
            return true;
        };
        if (!get_str("ss_filepath", out->ss_filepath, sizeof out->ss_filepath)) return SPADA_ERR_PARSE;
        if (!get_str("nn_filepath", out->nn_filepath, sizeof out->nn_filepath)) return SPADA_ERR_PARSE;
        if (!get_usize("pe_num", out->pe_num) || !get_usize("at_num", out->at_num) || !get_usize("lane_num", out->lane_num) ||
            !get_usize("cache_size", out->cache_size) || !get_usize("word_byte", out->word_byte) ||
            !get_usize("mem_latency", out->mem_latency) || !get_usize("cache_latency", out->cache_latency) ||
            !get_usize("channel", out->channel))
            return SPADA_ERR_PARSE;
        if (!get_f32("freq", out->freq) || !get_f32("bandwidth_per_channel", out->bandwidth_per_channel)) return SPADA_ERR_PARSE;
        {
            const JVal *v = need("block_shape", JVal::ARR);
            if (!v) return SPADA_ERR_PARSE;
            if (v->arr.size() != 2 || v->arr[0].t != JVal::NUM || v->arr[1].t != JVal::NUM || !v->arr[0].integral ||
                !v->arr[1].integral || v->arr[0].num < 0 || v->arr[1].num < 0)
                return fail(SPADA_ERR_PARSE, "%s: field `block_shape` must be an array of 2 non-negative integers", path);
            out->block_shape[0] = (uint64_t)v->arr[0].num;
            out->block_shape[1] = (uint64_t)v->arr[1].num;
        }
        // optional engine keys
        out->gpus = 1;
        out->accumulator = SPADA_ACC_LDS_HASH;
        out->repeat = 1;
        auto it = root.obj.find("gpus");
        if (it != root.obj.end() && it->second.t == JVal::NUM && it->second.num >= 1) out->gpus = (uint32_t)it->second.num;
        it = root.obj.find("repeat");
        if (it != root.obj.end() && it->second.t == JVal::NUM && it->second.num >= 1) out->repeat = (uint32_t)it->second.num;
        it = root.obj.find("accumulator");
        if (it != root.obj.end()) {
            if (it->second.t != JVal::STR || (it->second.str != "lds_hash" && it->second.str != "sort_merge"))
                return fail(SPADA_ERR_PARSE, "%s: field `accumulator` must be \"lds_hash\" or \"sort_merge\"", path);
            out->accumulator = it->second.str == "sort_merge" ? SPADA_ACC_SORT_MERGE : SPADA_ACC_LDS_HASH;
        }
        return SPADA_OK;
    });
}

}  // extern "C"
