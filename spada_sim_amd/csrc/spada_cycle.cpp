// spada_cycle.cpp -- the cycle-level Spada model behind include/spada_cycle.h.
//
// A restatement, in C++ and with this library's own data structures, of what the reference simulates cycle by cycle; every
// unit cites the reference lines whose behaviour it follows (paths relative to /root/reference/src).  Host code only.
//
//   Dram / PsumStore / FiberCache   storage.rs:150-323, :326-458, :460-1007   A / B memories, psum DRAM, the shared fiber cache
//   Planner                         scheduler.rs, rowwise_perf_adjust.rs      blocks, windows, merge tasks, block-height policy
//   ProcessingElement               simulator.rs:41-408                       stream buffers, multipliers, psum buffers with
//                                                                             tail flags, sorting network, merge tree
//   TreeMerger                      adder_tree.rs                             8-way comparator tree + adder
//   CycleModel                      simulator.rs:410-1251                     the cycle loop, result assembly, counters
//
// Deliberate differences (none changes a counter or a value):
//   * rows are indexed in place instead of cloned per access (storage.rs:785, :862 clone the whole fiber per request);
//   * every request is made with no_delay = true in the reference (simulator.rs:925, :940, :1223, :1235), which makes its
//     pending-request table a no-op; the table is not modelled;
//   * the policies that are constructed and fed but never selected (adjust_scheme is hard-wired to 3, scheduler.rs:203: the
//     energy trackers of rowwise_adjust.rs, colwise_*_adjust.rs, block_topo_tracker.rs) are not modelled;
//   * the progress lines the reference prints for every finished window (simulator.rs:563-573) are not printed;
//   * containers the reference iterates in hash order are visited in ascending key order (see spada_cycle.h);
//   * a window never reaches past its block (Planner::next_window: the reference's Op accelerator multiplies the elements of
//     a partial last row group several times), and the first block is fitted to the matrix for every accelerator.
// Where the reference would panic (an unwrap on a missing entry) this model throws and the C entry point reports an error.
#include "spada_cycle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <queue>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "spada_internal.hpp"

namespace {

constexpr uint64_t NONE = ~0ull;   // usize::MAX

struct Elem {   // storage.rs:22-32
    uint64_t row, col;
    double val;
};
constexpr Elem END_MARK{NONE, NONE, 0.0};   // "row drained" marker in the streams (simulator.rs:331, :1210)
inline bool is_end(const Elem &e) { return e.row == NONE && e.col == NONE; }

struct Slot {   // Option<Element>
    bool has = false;
    Elem e{0, 0, 0.0};
    static Slot of(const Elem &x) { return Slot{true, x}; }
};

[[noreturn]] void panic(const std::string &what) { throw std::runtime_error(what); }

// ---- fibers and memories -----------------------------------------------------------------------------------------------
struct Fiber {   // CsrRow, storage.rs:34-113
    uint64_t id = 0;
    std::vector<uint64_t> cols;
    std::vector<double> vals;
    uint64_t consumed = 0;
    uint64_t words() const { return 2 * (cols.size() - consumed); }   // size(): an element is two words
    uint64_t len() const { return cols.size(); }
    void append(const Fiber &o)
    {
        cols.insert(cols.end(), o.cols.begin(), o.cols.end());
        vals.insert(vals.end(), o.vals.begin(), o.vals.end());
    }
    void push(uint64_t c, double v)
    {
        cols.push_back(c);
        vals.push_back(v);
    }
};

struct Dram {   // CsrMatStorage, storage.rs:150-323 (arrays borrowed)
    spada_csr_view m{};
    const uint64_t *remap = nullptr;   // row_remap: new row -> raw row
    uint64_t read_words = 0, write_words = 0;
    uint64_t rows() const { return m.rows; }
    uint64_t raw(uint64_t r) const { return remap ? remap[r] : r; }
    uint64_t row_len(uint64_t r) const   // get_ele_num(r, r + 1)
    {
        const uint64_t q = raw(r);
        return m.indptr[q + 1] - m.indptr[q];
    }
    // read_scalars, storage.rs:279-323: up to `num` elements of row r from element offset `off`; the elements carry the raw row
    void read_scalars(uint64_t r, uint64_t off, uint64_t num, std::vector<Elem> &out)
    {
        out.clear();
        if (r >= rows()) panic("A: row out of range");
        if (num == 0) return;
        const uint64_t q = raw(r), s = m.indptr[q] + off, e = m.indptr[q + 1];
        if (s >= e) panic("A: read past the end of a row");
        for (uint64_t p = s; p < std::min(s + num, e); ++p) out.push_back(Elem{q, m.indices[p], m.data[p]});
        read_words += 2 * out.size();
    }
    // read_row, storage.rs:241-250 -> read, :163-194: the whole row as a fiber named r
    Fiber read_row(uint64_t r)
    {
        if (r >= rows()) panic("B: row out of range");
        const uint64_t q = raw(r), s = m.indptr[q], e = m.indptr[q + 1];
        Fiber f;
        f.id = r;
        f.cols.assign(m.indices + s, m.indices + e);
        f.vals.assign(m.data + s, m.data + e);
        read_words += f.words();
        return f;
    }
};

struct PsumStore {   // VectorStorage, storage.rs:326-458: the psum DRAM
    std::unordered_map<uint64_t, Fiber> data;
    uint64_t read_words = 0, write_words = 0;
    bool contains(uint64_t id) const { return data.count(id) != 0; }
    void write(const Fiber &f)   // :368-385: append to an existing fiber or insert
    {
        auto it = data.find(f.id);
        if (it == data.end()) data.emplace(f.id, f);
        else it->second.append(f);
        write_words += f.words();
    }
    const Fiber &read_row(uint64_t id)   // :398-412
    {
        auto it = data.find(id);
        if (it == data.end()) panic("psum DRAM: fiber " + std::to_string(id) + " not found");
        read_words += it->second.words();
        return it->second;
    }
    // consume_scalars, :414-440: elements [off, off + num) of the fiber; the fiber is dropped once its tail has been read
    bool consume_scalars(uint64_t id, uint64_t off, uint64_t num, std::vector<Elem> &out)
    {
        out.clear();
        auto it = data.find(id);
        if (it == data.end()) return false;
        const Fiber &f = it->second;
        const uint64_t t = std::min<uint64_t>(off + num, f.len());
        if (off > t) panic("psum DRAM: offset past the end of a fiber");
        for (uint64_t p = off; p < t; ++p) out.push_back(Elem{f.id, f.cols[p], f.vals[p]});
        read_words += 2 * (t - off);
        if (t == f.len()) data.erase(it);
        return true;
    }
    void write_element(uint64_t id, const Elem &e)   // :446-457
    {
        auto it = data.find(id);
        if (it == data.end()) {
            it = data.emplace(id, Fiber{}).first;
            it->second.id = id;
        }
        it->second.push(e.col, e.val);
        write_words += 2;
    }
};

// The fiber cache shared by all PEs and tree mergers (LatencyPriorityCache, storage.rs:460-1007): B rows and partial-sum fibers,
// capacity in words, eviction by the smallest "last A row that used this fiber" key; evicted psum fibers go to the psum DRAM.
struct FiberCache {
    uint64_t capacity = 0, used = 0;
    uint64_t read_words = 0, write_words = 0, miss_words = 0, b_evict_words = 0, psum_evict_words = 0;
    uint64_t b_words = 0, psum_words = 0;   // occupation by kind
    uint64_t psum_base = 0;                  // ids >= psum_base are psum fibers (is_psum_row, :672-674)
    std::unordered_map<uint64_t, Fiber> rows;   // rowmap
    std::set<uint64_t> psum_ids;                 // the psum fibers among them, ascending (see "ascending key order" above)
    using Key = std::pair<uint64_t, uint64_t>;   // (priority, id)
    std::priority_queue<Key, std::vector<Key>, std::greater<Key>> heap;
    std::unordered_map<uint64_t, uint64_t> live_priority;   // valid_pq_row_dict
    Dram *b = nullptr;
    PsumStore *psum = nullptr;
    uint64_t psum_membership = 0;   // bumped whenever a psum fiber enters or leaves `rows` (lets idle mergers skip re-scans)

    bool is_psum(uint64_t id) const { return id >= psum_base; }
    bool contains(uint64_t id) const { return rows.count(id) != 0; }

    void touch(uint64_t id, uint64_t priority)   // :569-574, :783-789: the live priority only grows
    {
        auto it = live_priority.find(id);
        if (it == live_priority.end()) it = live_priority.emplace(id, priority).first;
        else it->second = std::max(it->second, priority);
        heap.emplace(it->second, id);
    }

    // freeup_space, :591-649: evict until `need` more words fit; `pinned` is the fiber being written and is never a victim
    void make_room(uint64_t pinned, uint64_t need)
    {
        while (!heap.empty() && used + need > capacity) {
            uint64_t victim = NONE;
            if (b_words < need) {
                // no amount of B rows would do: take a psum fiber (:603-609 takes whichever the hash map yields first)
                for (uint64_t id : psum_ids)
                    if (id != pinned) {
                        victim = id;
                        break;
                    }
                if (victim == NONE) panic("fiber cache: no psum fiber left to evict");
            } else {
                for (;;) {   // smallest live (priority, id), skipping stale heap entries and the pinned fiber (:536-553)
                    std::vector<Key> held;
                    Key k{0, 0};
                    bool got = false;
                    while (!heap.empty()) {
                        k = heap.top();
                        heap.pop();
                        if (k.second == pinned) {
                            held.push_back(k);
                            continue;
                        }
                        got = true;
                        break;
                    }
                    for (const Key &h : held) heap.push(h);
                    if (!got) panic("fiber cache: nothing left to evict");
                    auto lp = live_priority.find(k.second);
                    if (lp != live_priority.end() && lp->second == k.first && rows.count(k.second)) {
                        victim = k.second;
                        break;
                    }
                }
            }
            auto it = rows.find(victim);
            const uint64_t w = it->second.words();
            used -= w;
            if (is_psum(victim)) {
                psum_evict_words += w;
                psum_words -= w;
                psum->write(it->second);
                ++psum_membership;
            } else {
                b_words -= w;
                b_evict_words += w;
            }
            if (is_psum(victim)) psum_ids.erase(victim);
            rows.erase(it);
        }
        if (used + need > capacity) panic("fiber cache: not enough space for " + std::to_string(need) + " words");
    }

    void insert(Fiber &&f, uint64_t priority, uint64_t id)   // write, :555-589
    {
        const uint64_t w = f.words();
        if (used + w > capacity) make_room(f.id, w);
        used += w;
        (is_psum(f.id) ? psum_words : b_words) += w;
        touch(id, priority);
        write_words += w;
        if (is_psum(id)) {
            ++psum_membership;
            psum_ids.insert(id);
        }
        rows[id] = std::move(f);
    }

    void swapout(uint64_t id)   // :651-670: a finished fiber leaves for the psum DRAM
    {
        auto it = rows.find(id);
        if (it == rows.end()) panic("fiber cache: swapout of a fiber that is not cached");
        const uint64_t w = it->second.words();
        used -= w;
        (is_psum(id) ? psum_words : b_words) -= w;
        psum->write(it->second);
        rows.erase(it);
        if (is_psum(id)) psum_ids.erase(id);
        ++psum_membership;
    }

    // append_psum_to, :676-735 (a chunk) and append_element_to, :918-985 (one element)
    void append(uint64_t id, const Fiber &chunk)
    {
        const uint64_t w = chunk.words();
        auto it = rows.find(id);
        if (it != rows.end()) {
            make_room(id, w);
            it = rows.find(id);
            used += w;
            (is_psum(id) ? psum_words : b_words) += w;
            write_words += w;
            it->second.append(chunk);
        } else if (psum->contains(id)) {
            psum->write(chunk);   // the fiber was spilled: the chunk follows it
        } else {
            touch(id, id);
            make_room(id, w);
            used += w;
            (is_psum(id) ? psum_words : b_words) += w;
            write_words += w;
            rows[id] = chunk;
            if (is_psum(id)) psum_ids.insert(id);
            ++psum_membership;
        }
    }
    void append_element(uint64_t id, const Elem &e)
    {
        auto it = rows.find(id);
        if (it != rows.end()) {
            make_room(id, 2);
            it = rows.find(id);
            used += 2;
            (is_psum(id) ? psum_words : b_words) += 2;
            write_words += 2;
            it->second.push(e.col, e.val);
        } else if (psum->contains(id)) {
            psum->write_element(id, e);
        } else {
            touch(id, id);
            make_room(id, 2);
            used += 2;
            (is_psum(id) ? psum_words : b_words) += 2;
            write_words += 2;
            Fiber f;
            f.id = id;
            f.push(e.col, e.val);
            rows[id] = std::move(f);
            if (is_psum(id)) psum_ids.insert(id);
            ++psum_membership;
        }
    }

    // request_read_scalars, :737-819.  loc = (A row, fiber id).  false: nothing was asked for (num == 0)
    bool read(uint64_t a_row, uint64_t id, uint64_t off, uint64_t num, std::vector<Elem> &out)
    {
        out.clear();
        if (num == 0) return false;
        auto it = rows.find(id);
        if (it != rows.end()) {
            if (off == 0) touch(id, a_row);
            const Fiber &f = it->second;
            const uint64_t t = std::min<uint64_t>(off + num, f.len());
            if (off > t) panic("fiber cache: offset past the end of a fiber");
            read_words += 2 * (t - off);
            for (uint64_t p = off; p < t; ++p) out.push_back(Elem{f.id, f.cols[p], f.vals[p]});
            return true;
        }
        Fiber f = is_psum(id) ? psum->read_row(id) : b->read_row(id);   // miss: the whole fiber is brought in
        miss_words += f.words();
        const uint64_t t = std::min<uint64_t>(off + num, f.len());
        if (off > t) panic("fiber cache: offset past the end of a fiber");
        for (uint64_t p = off; p < t; ++p) out.push_back(Elem{f.id, f.cols[p], f.vals[p]});
        insert(std::move(f), a_row, id);
        return true;
    }

    // request_consume_scalars, :821-916: like read, but the elements leave the cache (merge inputs are read once)
    bool consume(uint64_t id, uint64_t off, uint64_t num, std::vector<Elem> &out)
    {
        out.clear();
        if (num == 0) return false;
        auto it = rows.find(id);
        if (it != rows.end()) {
            Fiber &f = it->second;
            const uint64_t t = std::min<uint64_t>(off + num, f.len());
            if (off > t) panic("fiber cache: offset past the end of a fiber");
            const uint64_t w = 2 * (t - off);
            for (uint64_t p = off; p < t; ++p) out.push_back(Elem{f.id, f.cols[p], f.vals[p]});
            read_words += w;
            used -= w;
            (is_psum(id) ? psum_words : b_words) -= w;
            for (uint64_t k = 0; k < t - off; ++k) {   // rowmap_consume, :523-534
                if (f.words() == 0) panic("fiber cache: consume from an empty fiber");
                ++f.consumed;
            }
            if (f.words() == 0) {
                rows.erase(it);
                if (is_psum(id)) psum_ids.erase(id);
                ++psum_membership;
            }
            return true;
        }
        if (is_psum(id)) {
            if (psum->consume_scalars(id, off, num, out)) {
                read_words += 2 * out.size();
                miss_words += 2 * out.size();
            }
            return true;
        }
        // a B row streamed past the cache (:901-912)
        const uint64_t q = b->raw(id);
        if (id < b->rows() && off < b->m.indptr[q + 1] - b->m.indptr[q]) {
            b->read_scalars(id, off, num, out);
            for (Elem &e : out) e.row = id;
            read_words += 2 * out.size();
            miss_words += 2 * out.size();
        }
        return true;
    }
};

// ---- the planner: blocks, windows, merge tasks (scheduler.rs) ----------------------------------------------------------
struct Task {   // scheduler.rs:17-46
    uint64_t block = 0, window = 0, group = 1;
    bool merge = false;
    std::vector<Slot> a;   // one A scalar per lane; merge tasks: (row, fiber id) with value 1.0
    uint64_t traffic = 0, start_cycle = 0;
};

struct Block {   // BlockTracker, :67-100
    uint64_t anchor[2] = {0, 0}, shape[2] = {0, 0};
    bool is_merge = false;
    std::vector<uint64_t> assigned, total;   // a_cols_assigned, a_cols_num per block row
    std::vector<uint64_t> windows;
    std::vector<char> is_tail;
    uint64_t latency = 0;                     // RowwiseLatencyBlockInfo.latency, rowwise_perf_adjust.rs:79-92
};

struct Window {   // WindowTracker, :102-135
    uint64_t anchor[2] = {0, 0}, block = 0, shape[2] = {0, 0};
    std::vector<uint64_t> streamed;                       // b_cols_assigned per lane
    std::vector<uint64_t> lane_len;                       // length of the lane's fiber (fixed once the window exists: B rows, or
                                                          // partial fibers whose producing window has retired)
    std::vector<std::pair<uint64_t, uint64_t>> lane_src;  // lane2idx: (A row, fiber id) or (NONE, NONE)
    std::vector<std::pair<uint64_t, uint64_t>> outputs;   // arow_addr_pairs: per group (C row, psum address)
};

// Block-height policy of the Spada accelerator (adjust_scheme 3): rowwise_perf_adjust.rs
struct HeightPolicy {
    struct Group {
        uint64_t lo = 0, hi = 0;
        std::map<uint64_t, std::pair<uint64_t, uint64_t>> cost;   // block height -> (latency, A elements)
    };
    std::vector<Group> groups;
    std::vector<uint64_t> group_of;   // rgmap
    uint64_t cur = NONE, fixed = NONE, lanes = 8;
    std::vector<uint64_t> sample_bounds;

    // parse_group, :36-77: consecutive rows whose lengths stay within a factor 1.5 of their predecessor
    void parse(const Dram &a, float var)
    {
        const uint64_t n = a.rows();
        group_of.assign(n, 0);
        uint64_t prev = NONE, lo = 0;
        auto close = [&](uint64_t hi) {
            Group g;
            g.lo = lo;
            g.hi = hi;
            groups.push_back(g);
            for (uint64_t r = lo; r < hi; ++r) group_of[r] = groups.size() - 1;
        };
        for (uint64_t i = 0; i <= n; ++i) {
            if (i == n) {
                close(i);
            } else {
                const uint64_t len = a.row_len(i);
                if (len == 0) continue;
                if (prev == NONE) {
                    prev = len;
                } else if ((float)prev * var < (float)len || (float)prev > var * (float)len) {
                    close(i);
                    prev = len;
                    lo = i;
                } else {
                    prev = len;
                }
            }
        }
    }

    static float per_element(const std::pair<uint64_t, uint64_t> &c) { return (float)c.first / ((float)c.second + 0.0001f); }

    // adjust_block_shape, :121-231: height of the block that starts at row_s, given the height of the block before it
    uint64_t height(uint64_t row_s, uint64_t prev_height)
    {
        constexpr uint64_t WIDE = 128, SAMPLES = 4;
        if (group_of[row_s] != cur || !groups[cur].cost.count(prev_height)) {
            // a new group (or a height without measurements): start at 1; wide groups get a sampling schedule 1, 2, 4, 8
            cur = group_of[row_s];
            const Group &g = groups[cur];
            if (g.hi - g.lo > WIDE) {
                uint64_t row = row_s + 1;
                sample_bounds.clear();
                for (uint64_t i = 1; i <= lanes; i *= 2) {
                    row += SAMPLES * i;
                    sample_bounds.push_back(row);
                }
            }
            fixed = NONE;
            return 1;
        }
        Group &g = groups[cur];
        uint64_t h = 1;
        if (g.hi - g.lo > WIDE) {
            if (sample_bounds.empty()) panic("height policy: no sampling schedule");
            if (row_s >= sample_bounds.back()) {
                if (fixed == NONE) {   // sampling over: the height with the smallest latency per A element
                    float best = 3.4028235e38f;
                    uint64_t cand = 1;
                    while (cand <= lanes) {
                        auto it = g.cost.find(cand);
                        if (it != g.cost.end()) {
                            const float c = per_element(it->second);
                            if (c < best) {
                                best = c;
                                fixed = cand;
                            }
                        } else {
                            g.cost[cand] = {0, 0};
                            fixed = cand;
                            break;
                        }
                        cand *= 2;
                    }
                }
                h = fixed;
            } else {
                // the k-th sampling interval runs blocks of height 2^k
                const auto it = std::lower_bound(sample_bounds.begin(), sample_bounds.end(), row_s);
                const uint64_t idx = (uint64_t)(it - sample_bounds.begin());
                h = (it != sample_bounds.end() && *it == row_s) ? (1ull << (idx + 1)) : (1ull << idx);
            }
        } else {
            // narrow groups: keep doubling while that lowers the latency per element, then settle for the best seen
            const auto cur_c = g.cost.find(prev_height);
            const auto half_c = g.cost.find(prev_height / 2);
            bool grow = false;
            if (fixed == NONE) {
                if (half_c == g.cost.end()) grow = true;
                else {
                    // (:200-205 divides BOTH latencies by the A elements of the current height)
                    const float den = (float)cur_c->second.second + 0.0001f;
                    grow = (float)cur_c->second.first / den < (float)half_c->second.first / den;
                }
            }
            if (grow) {
                h = 2;   // (:207: block_row_num starts at 1 and is doubled once)
            } else {
                float best = 3.4028235e38f;
                for (const auto &kv : g.cost) {
                    const float c = per_element(kv.second);
                    if (c < best) {
                        best = c;
                        fixed = kv.first;
                    }
                }
                h = fixed;
            }
        }
        while (h > 1 && row_s + h >= g.hi) h /= 2;
        return h;
    }

    void record(uint64_t first_row, uint64_t height_, uint64_t latency, uint64_t a_elems)   // update_group_cost, :233-248
    {
        auto &c = groups[group_of[first_row]].cost[height_];
        c.first += latency;
        c.second += a_elems;
    }
};

struct Planner {
    // configuration
    uint64_t lanes = 8, n_rows = 0, psum_base = 0;
    int accel = SPADA_ACCEL_SPADA;
    uint64_t cache_latency = 0;
    uint64_t shape[2] = {1, 1};   // current block shape
    // cursor over A
    bool traversed = false;
    uint64_t row_s = NONE, col_s = NONE;
    std::vector<uint64_t> a_len, a_assigned;
    // fibers: length of B rows and of psum fibers (b_row_lens)
    std::vector<uint64_t> b_len;
    std::unordered_map<uint64_t, uint64_t> psum_len;
    // trackers
    std::vector<Block> blocks;
    std::unordered_map<uint64_t, Window> windows;
    std::map<uint64_t, std::vector<uint64_t>> outputs;   // output_tracker: C row -> its partial fibers
    std::set<uint64_t> tail_done;                         // a_tail_produced
    std::unordered_map<uint64_t, uint64_t> finished;      // a_row_finished: C row -> final fiber
    std::unordered_map<uint64_t, uint64_t> pending;       // row_rgstr_task: tasks in flight per C row
    uint64_t next_addr = 0, next_window_id = 0;
    uint64_t latest_block = NONE;
    HeightPolicy policy;
    uint64_t n_pe_merges = 0, n_tree_merges = 0, n_windows = 0;
    uint64_t fibers_changed = 0;   // bumped whenever `outputs`, `tail_done` or `pending` change

    uint64_t fiber_len(uint64_t id) const
    {
        if (id < psum_base) {
            if (id >= b_len.size()) panic("planner: B row " + std::to_string(id) + " out of range");
            return b_len[id];
        }
        auto it = psum_len.find(id);
        if (it == psum_len.end()) panic("planner: unknown fiber " + std::to_string(id));
        return it->second;
    }
    bool fiber_known(uint64_t id) const { return id < psum_base ? id < b_len.size() : psum_len.count(id) != 0; }

    bool block_done(uint64_t b) const   // is_block_finished, :276-288
    {
        const Block &k = blocks[b];
        for (size_t i = 0; i < k.total.size(); ++i)
            if (k.assigned[i] < k.total[i]) return false;
        return true;
    }
    static bool window_done(const Window &win)   // is_window_finished, :645-666: every lane has streamed its whole fiber
    {
        for (size_t l = 0; l < win.lane_len.size(); ++l)
            if (win.streamed[l] < win.lane_len[l]) return false;
        return true;
    }
    void seal(Window &w) const   // lane lengths (b_row_lens at the time the reference would look them up: they no longer change)
    {
        w.lane_len.assign(w.lane_src.size(), 0);
        for (size_t l = 0; l < w.lane_src.size(); ++l)
            if (w.lane_src[l].second != NONE) w.lane_len[l] = fiber_len(w.lane_src[l].second);
    }
    void mark_tail_rows(uint64_t b)   // label_finished_rows, :290-300
    {
        const Block &k = blocks[b];
        if (k.is_merge) return;
        for (size_t o = 0; o < k.is_tail.size(); ++o)
            if (k.is_tail[o] && !finished.count(k.anchor[0] + o) && tail_done.insert(k.anchor[0] + o).second) ++fibers_changed;
    }

    void fit_height()   // adjust_block_row, :668-706
    {
        if (accel == SPADA_ACCEL_SPADA) shape[0] = policy.height(row_s, shape[0]);
        else
            while (row_s + shape[0] > n_rows) shape[0] = std::max<uint64_t>(1, shape[0] / 2);
    }
    bool nothing_left_at(uint64_t r0, uint64_t c0) const   // is_block_valid, :608-620
    {
        for (uint64_t r = r0; r < r0 + shape[0]; ++r) {
            if (r >= n_rows || c0 >= a_len[r]) continue;
            return c0 < a_assigned[r];
        }
        return true;
    }
    uint64_t open_block()   // the block at (row_s, col_s) with the current shape: :306-330 / :342-366, set_block :755-800
    {
        Block k;
        k.anchor[0] = row_s;
        k.anchor[1] = col_s;
        k.shape[0] = shape[0];
        k.shape[1] = shape[1];
        for (uint64_t o = 0; o < shape[0]; ++o) {
            const uint64_t r = row_s + o;
            if (r >= n_rows) panic("planner: block reaches past the last row of A");
            const uint64_t len = a_len[r], end = col_s + shape[1] < col_s ? NONE : col_s + shape[1];
            k.total.push_back(std::max(std::min(len, end), col_s) - col_s);
            k.is_tail.push_back(end >= len);
            a_assigned[r] += k.total.back();
        }
        k.assigned.assign(shape[0], 0);
        blocks.push_back(std::move(k));
        col_s = col_s + shape[1] < col_s ? NONE : col_s + shape[1];
        return blocks.size() - 1;
    }
    uint64_t next_block()   // :302-379; NONE when A is exhausted
    {
        for (;;) {
            if (row_s == NONE && col_s == NONE) {
                row_s = col_s = 0;
                if (n_rows == 0) return NONE;
                fit_height();   // (the reference adjusts only Spada's first block, :304: an Op run on fewer rows than lanes indexes
                                // past the end of A and panics; the other accelerators' first block is fitted here as well)
                return open_block();
            }
            if (row_s >= n_rows) return NONE;
            if (!nothing_left_at(row_s, col_s)) {
                if (accel != SPADA_ACCEL_SPADA)   // adjust_block_col, :708-731 (scheme 3 keeps the shape)
                    while (row_s + shape[0] > n_rows) shape[0] = std::max<uint64_t>(1, shape[0] / 2);
                return open_block();
            }
            row_s += shape[0];
            if (row_s < n_rows) {
                col_s = a_assigned[row_s];
                fit_height();
            } else {
                col_s = 0;
            }
        }
    }

    // next_window, :482-606: the next [h, lanes / h] window of the block, sliding along K; false when the block has none left
    bool next_window(uint64_t b, Dram &a, uint64_t cycle, uint64_t *latency, Task *task)
    {
        Block &k = blocks[b];
        uint64_t wshape[2], anchor[2];
        const uint64_t id = next_window_id++;   // (a token is taken even when no window follows, :497, :508)
        if (k.windows.empty()) {
            wshape[0] = accel == SPADA_ACCEL_SPADA ? k.shape[0] : shape[0];   // adjust_window, :733-753
            // (the reference makes the window lanes / height wide even when the block is narrower -- the Op accelerator's blocks
            // are one column wide, and once the last row group has forced the height below lane_num its windows would take the
            // same A elements again for every following block: a wrong product.  The window is kept inside its block here.)
            wshape[1] = std::min(lanes / wshape[0], k.shape[1]);
            anchor[0] = k.anchor[0];
            anchor[1] = k.anchor[1];
            *latency = cache_latency;
        } else {
            const Window &p = windows.at(k.windows.back());
            wshape[0] = p.shape[0];
            wshape[1] = p.shape[1];
            anchor[0] = p.anchor[0];
            anchor[1] = p.anchor[1];
            const uint64_t row_lim = k.anchor[0] + k.shape[0];
            const uint64_t widest = *std::max_element(k.total.begin(), k.total.end());
            const uint64_t col_lim = k.anchor[1] + std::min(k.shape[1], widest);
            if (anchor[0] >= row_lim) return false;
            if (anchor[1] + wshape[1] < col_lim) {
                anchor[1] += wshape[1];
            } else {
                while (anchor[0] < row_lim) {   // next rows of the block that still have elements (:526-540)
                    anchor[1] = k.anchor[1];
                    anchor[0] += wshape[0];
                    bool empty = true;   // is_window_valid, :622-643
                    for (uint64_t r = anchor[0]; r < std::min(anchor[0] + wshape[0], row_lim); ++r)
                        if (r < n_rows && anchor[1] < a_len[r]) {
                            empty = false;
                            break;
                        }
                    if (!empty) break;
                }
                if (anchor[0] >= row_lim) return false;
            }
            *latency = 0;
        }
        Window w;
        w.anchor[0] = anchor[0];
        w.anchor[1] = anchor[1];
        w.block = b;
        w.shape[0] = wshape[0];
        w.shape[1] = wshape[1];
        for (uint64_t o = 0; o < wshape[0]; ++o) w.outputs.emplace_back(anchor[0] + o, next_addr++);
        Task t;
        t.block = b;
        t.window = id;
        t.group = wshape[1];
        t.merge = false;
        t.start_cycle = cycle;
        std::vector<Elem> got;
        for (uint64_t r = anchor[0]; r < anchor[0] + wshape[0]; ++r) {
            if (r >= n_rows) panic("planner: window reaches past the last row of A");
            const uint64_t num = std::min(std::max(a_len[r], anchor[1]), anchor[1] + wshape[1]) - anchor[1];
            a.read_scalars(r, anchor[1], num, got);
            k.assigned[r - k.anchor[0]] += got.size();
            for (const Elem &e : got) {
                w.lane_src.emplace_back(e.row, e.col);
                t.a.push_back(Slot::of(Elem{id, e.col, e.val}));   // the row field carries the window token (:566)
            }
            for (uint64_t pad = got.size(); pad < wshape[1]; ++pad) {
                w.lane_src.emplace_back(NONE, NONE);
                t.a.push_back(Slot{});
            }
        }
        for (const auto &o : w.outputs) ++pending[o.first];
        ++fibers_changed;
        while (w.lane_src.size() < lanes) w.lane_src.emplace_back(NONE, NONE);   // (only when the window was narrowed, above)
        w.streamed.assign(w.lane_src.size(), 0);
        seal(w);
        windows.emplace(id, std::move(w));
        k.windows.push_back(id);
        ++n_windows;
        *task = std::move(t);
        return true;
    }

    // merge_task, :381-480: up to lanes / 2 pairs of partial fibers, each pair merged by one lane pair of a PE
    bool pe_merge_task(uint64_t cycle, Task *task)
    {
        uint64_t pairs = 0;
        for (const auto &kv : outputs) {
            if (pairs >= lanes / 2) break;
            pairs += kv.second.size() / 2;
        }
        if ((traversed && pairs == 0) || (!traversed && pairs < lanes / 2)) return false;
        std::vector<std::pair<uint64_t, uint64_t>> picked;   // (C row, fiber)
        for (auto &kv : outputs)
            while (kv.second.size() > 1 && picked.size() < lanes) {
                picked.emplace_back(kv.first, kv.second[0]);
                picked.emplace_back(kv.first, kv.second[1]);
                kv.second.erase(kv.second.begin(), kv.second.begin() + 2);
                ++fibers_changed;
            }
        const uint64_t b = blocks.size(), wid = next_window_id++;
        Window w;
        w.block = b;
        w.shape[0] = lanes / 2;
        w.shape[1] = 2;
        Task t;
        t.block = b;
        t.window = wid;
        t.group = 2;
        t.merge = true;
        t.start_cycle = cycle;
        Block k;
        k.shape[0] = lanes / 2;
        k.shape[1] = 2;
        k.is_merge = true;
        for (uint64_t g = 0; g < lanes / 2; ++g) {
            if (g < picked.size() / 2) {
                w.outputs.emplace_back(picked[2 * g].first, next_addr++);
                for (int s = 0; s < 2; ++s) {
                    const auto &p = picked[2 * g + s];
                    t.a.push_back(Slot::of(Elem{p.first, p.second, 1.0}));
                    w.lane_src.push_back(p);
                }
                k.total.push_back(2);
                k.assigned.push_back(2);
                ++pending[picked[2 * g].first];
            } else {
                w.outputs.emplace_back(NONE, next_addr++);   // (an address is taken for unused groups too, :427)
                for (int s = 0; s < 2; ++s) {
                    t.a.push_back(Slot{});
                    w.lane_src.emplace_back(NONE, NONE);
                }
                k.total.push_back(0);
                k.assigned.push_back(0);
            }
            k.is_tail.push_back(0);
        }
        w.streamed.assign(lanes, 0);
        seal(w);
        k.windows.push_back(wid);
        blocks.push_back(std::move(k));
        windows.emplace(wid, std::move(w));
        ++n_pe_merges;
        *task = std::move(t);
        return true;
    }

    // in_cache_merge_task, :820-920: up to `width` cached partial fibers of ONE C row for a tree merger.  Rows with at least
    // `width` fibers first; otherwise a row whose last block has been issued and that still has more than one fiber.
    bool tree_merge_task(uint64_t width, const FiberCache &cache, uint64_t cycle, Task *task)
    {
        std::vector<std::pair<uint64_t, uint64_t>> picked;
        auto all_cached = [&](const std::vector<uint64_t> &v) {
            for (uint64_t f : v)
                if (!cache.contains(f)) return false;
            return true;
        };
        auto take = [&](uint64_t row, std::vector<uint64_t> &v) {
            const size_t n = std::min<size_t>(width, v.size());
            for (size_t i = 0; i < n; ++i) picked.emplace_back(row, v[i]);
            v.erase(v.begin(), v.begin() + n);
            ++fibers_changed;
        };
        for (auto &kv : outputs)
            if (kv.second.size() >= width && all_cached(kv.second)) {
                take(kv.first, kv.second);
                break;
            }
        if (picked.empty())
            for (uint64_t row : tail_done) {
                auto it = outputs.find(row);
                if (it == outputs.end()) continue;
                if (it->second.size() > 1 && all_cached(it->second)) {
                    take(row, it->second);
                    break;
                }
            }
        if (picked.empty()) return false;
        const uint64_t b = blocks.size(), wid = next_window_id++;
        Window w;
        w.block = b;
        w.shape[0] = 1;
        w.shape[1] = width;
        w.outputs.emplace_back(picked[0].first, next_addr++);
        Task t;
        t.block = b;
        t.window = wid;
        t.group = width;
        t.merge = true;
        t.start_cycle = cycle;
        for (const auto &p : picked) {
            t.a.push_back(Slot::of(Elem{p.first, p.second, 1.0}));
            w.lane_src.push_back(p);
        }
        for (size_t l = picked.size(); l < width; ++l) {
            t.a.push_back(Slot{});
            w.lane_src.emplace_back(NONE, NONE);
        }
        w.streamed.assign(width, 0);
        seal(w);
        ++pending[picked[0].first];
        Block k;
        k.shape[0] = 1;
        k.shape[1] = width;
        k.is_merge = true;
        k.total.push_back(picked.size());
        k.assigned.push_back(width);
        k.is_tail.push_back(0);
        k.windows.push_back(wid);
        blocks.push_back(std::move(k));
        windows.emplace(wid, std::move(w));
        ++n_tree_merges;
        *task = std::move(t);
        return true;
    }

    // assign_task, :242-274: the PE continues its block, joins the latest unfinished block, opens the next one, or -- once A
    // is exhausted -- merges pairs of partial fibers
    bool assign(bool has_task, uint64_t cur_block, Dram &a, uint64_t cycle, uint64_t *latency, Task *task)
    {
        *latency = 0;
        if (!has_task || block_done(cur_block)) {
            if (latest_block != NONE && !block_done(latest_block)) return next_window(latest_block, a, cycle, latency, task);
            const uint64_t b = next_block();
            if (b == NONE) {
                traversed = true;
                return pe_merge_task(cycle, task);
            }
            const bool got = next_window(b, a, cycle, latency, task);
            latest_block = b;
            return got;
        }
        return next_window(cur_block, a, cycle, latency, task);
    }

    // the bookkeeping when a unit has finished the window `w` (simulator.rs:640-662, :1100-1122)
    void retire_window(uint64_t w)
    {
        for (const auto &o : windows.at(w).outputs) {
            auto p = pending.find(o.first);
            if (p != pending.end()) --p->second;
            ++fibers_changed;
            if (fiber_known(o.second)) {
                auto &v = outputs[o.first];
                if (std::find(v.begin(), v.end(), o.second) == v.end()) v.push_back(o.second);
                ++fibers_changed;
            }
        }
    }
};

// ---- the PE datapath (simulator.rs:41-408) -----------------------------------------------------------------------------
struct Staged {   // a batch inside the sorting network / merge tree with its remaining latency
    std::vector<std::vector<Elem>> groups;
    uint64_t wait = 0;
};

struct ProcessingElement {
    uint64_t lanes = 8, sb_size = 4, pb_size = 8, pops = 2, sn_latency = 4, mt_latency = 4;
    std::vector<std::deque<Elem>> stream, psum;
    std::vector<Slot> a, bcur, prod;   // multiplier array: A scalars, current B operands, products of the last cycle
    std::vector<char> row_drained, sb_drained, full;
    std::vector<uint64_t> tail;
    std::deque<Staged> sorter, merger;
    bool has_task = false, config_unchanged = false;
    Task task;
    uint64_t mem_finish = NONE, drain_since = NONE;
    uint64_t group_lanes = 8;   // sorting_network.group_lane_num
    Window *win = nullptr;      // the task's window (node of an unordered_map: the address is stable)
    std::vector<Slot> bs_buf, prods_buf;   // per-cycle scratch (kept to avoid allocations)
    std::vector<Staged> spare;             // retired batches, reused for their buffers

    void init(uint64_t lanes_)
    {
        lanes = lanes_;
        stream.assign(lanes, {});
        psum.assign(lanes, {});
        a.assign(lanes, Slot{});
        bcur.assign(lanes, Slot{});
        prod.assign(lanes, Slot{});
        row_drained.assign(lanes, 0);
        sb_drained.assign(lanes, 1);
        full.assign(lanes, 0);
        tail.assign(lanes, 0);
        group_lanes = lanes;
    }
    bool lane_empty(uint64_t l) const { return !a[l].has || row_drained[l]; }   // MultiplierArray::is_empty, :112-114
    bool idle() const   // :293-307
    {
        for (uint64_t l = 0; l < lanes; ++l)
            if (!stream[l].empty() || !psum[l].empty() || !lane_empty(l)) return false;
        return sorter.empty() && merger.empty();
    }
    void set_a(const std::vector<Slot> &v)   // set_as, :61-70
    {
        for (uint64_t l = 0; l < lanes; ++l) {
            a[l] = l < v.size() ? v[l] : Slot{};
            row_drained[l] = !a[l].has;
        }
    }
    uint64_t set_task(bool got, uint64_t latency, Task &&t)   // :373-407
    {
        if (!got) {
            set_a({});
            mem_finish = NONE;
            has_task = false;
            drain_since = NONE;
            config_unchanged = false;
            return 0;
        }
        config_unchanged = has_task && task.group == t.group;
        task = std::move(t);
        has_task = true;
        group_lanes = task.group;
        for (uint64_t l = 0; l < lanes; ++l) sb_drained[l] = !(l < task.a.size() && task.a[l].has);
        set_a(task.a);
        mem_finish = NONE;
        drain_since = NONE;
        return latency;
    }
    void feed(uint64_t l, bool more, const std::vector<Elem> &es)   // push_stream_buffer, :326-337
    {
        if (more) {
            for (const Elem &e : es) stream[l].push_back(e);
        } else if (!sb_drained[l]) {
            stream[l].push_back(END_MARK);
            sb_drained[l] = 1;
        }
    }
    // pop_stream_buffer, :339-363: lanes of a group of two or more share their streams pairwise -- the smaller head of the
    // pair's two streams goes first (its last element stays for its own lane), so a group's products leave in column order
    Slot take_b(uint64_t l)
    {
        if (!has_task || full[l]) return Slot{};
        auto pop = [&](uint64_t q) {
            if (stream[q].empty()) return Slot{};
            const Slot s = Slot::of(stream[q].front());
            stream[q].pop_front();
            return s;
        };
        if (task.group < 2) return pop(l);
        const uint64_t left = (l / 2) * 2, right = left + 1;
        int first = -1;   // merge_idx(.., 1), :17-39: which head is smaller (ties: left)
        if (!stream[left].empty() && !stream[right].empty()) first = stream[left].front().col <= stream[right].front().col ? 0 : 1;
        else if (!stream[left].empty()) first = 0;
        else if (!stream[right].empty()) first = 1;
        if (first == 0 && stream[left].size() > 1) return pop(left);
        if (first == 1 && stream[right].size() > 1) return pop(right);
        return pop(l);
    }
    // MultiplierArray::set_bs + multiply, :72-111: the product of lane l is that of ITS b with the A scalar of its group whose
    // column is the b's row
    void multiply(const std::vector<Slot> &bs)
    {
        for (uint64_t l = 0; l < lanes; ++l) {
            if (bs[l].has && is_end(bs[l].e)) {
                row_drained[l] = 1;
                bcur[l] = Slot{};
            } else {
                bcur[l] = bs[l];
            }
        }
        const uint64_t g = task.group ? task.group : 1;
        for (uint64_t l = 0; l < lanes; ++l) {
            if (!bcur[l].has) {
                prod[l] = Slot{};
                continue;
            }
            const uint64_t g0 = (l / g) * g;
            bool matched = false;
            for (uint64_t q = g0; q < std::min(g0 + g, lanes); ++q)
                if (a[q].has && a[q].e.col == bcur[l].e.row) {
                    prod[l] = Slot::of(Elem{a[q].e.row, bcur[l].e.col, a[q].e.val * bcur[l].e.val});   // :100-101
                    matched = true;
                    break;
                }
            if (!matched) panic("PE: no A scalar matches a streamed B element");
        }
    }
    // update_tail_flags, :309-324: a group may release products below the smallest column any of its lanes can still produce
    void update_tails()
    {
        const uint64_t g = task.group ? task.group : 1;
        for (uint64_t s = 0; s < lanes; s += g) {
            uint64_t t = NONE;
            for (uint64_t l = s; l < std::min(s + g, lanes); ++l) {
                if (psum[l].size() >= 3) t = std::min(t, psum[l][2].col);
                else if (!lane_empty(l) && bcur[l].has) t = std::min(t, bcur[l].e.col);
                else if (!lane_empty(l)) t = 0;   // (:317 unwraps the operand; a lane without one holds its group back)
            }
            for (uint64_t l = s; l < std::min(s + g, lanes); ++l) tail[l] = t;
        }
    }
    // one call per cycle: SortingNetwork::pop_elements, :143-171 and MergeTree::pop_elements, :199-230
    static bool release(std::deque<Staged> &q, Staged *out)
    {
        bool got = false;
        for (auto it = q.begin(); it != q.end(); ++it)
            if (it->wait == 0) {
                *out = std::move(*it);
                q.erase(it);
                got = true;
                break;
            }
        for (Staged &s : q)
            if (s.wait) --s.wait;
        return got;
    }
};

// ---- the tree merger (adder_tree.rs) -----------------------------------------------------------------------------------
struct TreeMerger {
    uint64_t width = 8, depth = 3;
    std::vector<std::vector<Slot>> node;       // level 0 = root
    std::vector<std::vector<char>> drained;
    std::vector<Elem> scalars;                 // the multiplier's A operands (value 1.0)
    bool has_scalars = false;
    Slot b, c, acc;                            // multiplier operand / product, the adder's running element
    bool has_task = false;
    Task task;
    Window *win = nullptr;
    uint64_t failed_at[2] = {NONE, NONE};   // (planner, cache) change counters of the last scan that found nothing to merge

    void init(uint64_t w)
    {
        width = w;
        node.clear();
        drained.clear();
        for (uint64_t lw = 1; lw <= w; lw *= 2) {
            node.emplace_back(lw);
            drained.emplace_back(lw, 0);
        }
        depth = node.size() - 1;
    }
    bool idle() const   // :247-250
    {
        for (const auto &lvl : node)
            for (const Slot &s : lvl)
                if (s.has && !is_end(s.e)) return false;
        return (!has_scalars || !b.has) && !acc.has;
    }
    void set_task(bool got, Task &&t)   // :252-268
    {
        if (!got) {
            has_scalars = false;
            has_task = false;
            return;
        }
        task = std::move(t);
        has_task = true;
        for (auto &lvl : drained) std::fill(lvl.begin(), lvl.end(), 0);
        scalars.clear();
        for (const Slot &s : task.a)
            if (s.has) scalars.push_back(s.e);
        has_scalars = true;
    }
    void push_leaf(uint64_t leaf, const Elem &e)   // :127-143
    {
        if (is_end(e)) {
            drained[depth][leaf] = 1;
            return;
        }
        if (node[depth][leaf].has) panic("tree merger: leaf already occupied");
        node[depth][leaf] = Slot::of(e);
    }
    // MergeTree::update, :145-188: the root leaves; every empty inner node takes the smaller of its children once both are
    // ready (holding an element, or drained) -- ties go left; levels are visited top down, so an element climbs one level a cycle
    Slot step()
    {
        const Slot out = node[0][0];
        node[0][0] = Slot{};
        for (uint64_t lvl = 1; lvl <= depth; ++lvl)
            for (uint64_t left = 0; left < node[lvl].size(); left += 2) {
                Slot &parent = node[lvl - 1][left / 2];
                if (parent.has) continue;
                const uint64_t right = left + 1;
                Slot &le = node[lvl][left], &re = node[lvl][right];
                if ((drained[lvl][right] || re.has) && (drained[lvl][left] || le.has)) {
                    if (!re.has || (le.has && le.e.col <= re.e.col)) {
                        parent = le;
                        le = Slot{};
                    } else {
                        parent = re;
                        re = Slot{};
                    }
                }
                if (drained[lvl][right] && drained[lvl][left] && !le.has && !re.has) drained[lvl - 1][left / 2] = 1;
            }
        return out;
    }
    void multiply()   // Multiplier::multiply, :37-57
    {
        if (!has_scalars || !b.has) {
            c = Slot{};
            return;
        }
        for (const Elem &s : scalars)
            if (s.col == b.e.row) {
                c = Slot::of(Elem{s.row, b.e.col, s.val * b.e.val});
                return;
            }
        panic("tree merger: no scalar matches a streamed element");
    }
    Slot add(const Slot &in)   // Adder::add, :73-83: equal (row, column) accumulate, anything else pushes the held element out
    {
        if (acc.has && in.has && acc.e.row == in.e.row && acc.e.col == in.e.col) {
            acc.e.val += in.e.val;
            return Slot{};
        }
        const Slot out = acc;
        acc = in;
        return out;
    }
};

}  // namespace

// ---- the model ---------------------------------------------------------------------------------------------------------
struct spada_cycle_model {
    spada_cycle_config cfg{};
    Dram a, b;
    PsumStore psum;
    FiberCache cache;
    Planner plan;
    std::vector<ProcessingElement> pes;
    std::vector<TreeMerger> trees;
    std::vector<uint64_t> a_pending, drain_cycles;
    uint64_t cycle = 0;
    float words_per_cycle_channel = 1.0f;   // word_cycle_chan_bw, simulator.rs:457
    bool executed = false;
    // result
    std::vector<uint64_t> c_indptr, c_indices;
    std::vector<double> c_data;
    std::vector<Elem> scratch;
    Fiber chunk_buf;

    float pe_bandwidth() const { return words_per_cycle_channel * (float)cfg.channel / (float)cfg.pe_num; }

    // swapout_finished_psums, simulator.rs:985-1006: a row is complete when its last block has been issued, no task is
    // registered on it and exactly one fiber is left
    uint64_t settled_at = NONE;   // plan.fibers_changed after the last pass (nothing it looks at has changed since: skip)
    void settle_finished_rows()
    {
        if (settled_at == plan.fibers_changed) return;
        for (auto it = plan.tail_done.begin(); it != plan.tail_done.end();) {
            const uint64_t row = *it;
            auto p = plan.pending.find(row);
            auto o = plan.outputs.find(row);
            const bool quiet = p == plan.pending.end() || p->second == 0;
            const bool single = o == plan.outputs.end() || o->second.size() == 1;
            if (!(quiet && single)) {
                ++it;
                continue;
            }
            it = plan.tail_done.erase(it);
            ++plan.fibers_changed;
            if (o != plan.outputs.end()) {
                const uint64_t fiber = o->second[0];
                plan.finished[row] = fiber;
                plan.outputs.erase(o);
                if (cache.contains(fiber)) cache.swapout(fiber);
            }
        }
        settled_at = plan.fibers_changed;
    }

    // stream_b_row, simulator.rs:892-953.  Returns false when the lane's fiber has been streamed to its end.
    bool stream_lane(ProcessingElement &pe, uint64_t lane, uint64_t room, std::vector<Elem> &out)
    {
        out.clear();
        if (!pe.has_task) return false;
        Window &w = *pe.win;
        const auto src = w.lane_src[lane];
        if (src.second == NONE) return false;
        const uint64_t off = w.streamed[lane];
        if (!cache.contains(src.second) && off == 0)
            pe.task.traffic += (uint64_t)((float)cfg.mem_latency * words_per_cycle_channel);   // :920-923
        const bool asked = pe.task.merge ? cache.consume(src.second, off, room, out) : cache.read(src.first, src.second, off, room, out);
        if (!asked) return true;   // nothing was asked for (the buffer is full): not drained
        if (out.empty()) return false;
        w.streamed[lane] += out.size();
        return true;
    }

    void write_psums(ProcessingElement &pe, const Staged &merged)   // simulator.rs:955-983
    {
        if (!pe.has_task) return;
        const Window &w = *pe.win;
        for (size_t g = 0; g < merged.groups.size(); ++g) {
            const auto &es = merged.groups[g];
            if (es.empty()) continue;
            if (g >= w.outputs.size()) panic("PE: more groups than outputs");
            Fiber &f = chunk_buf;
            f.id = w.outputs[g].second;
            f.cols.clear();
            f.vals.clear();
            f.consumed = 0;
            for (const Elem &e : es) f.push(e.col, e.val);
            plan.psum_len[f.id] += f.len();
            cache.append(f.id, f);
        }
    }

    void pe_cycle(uint64_t p)   // one PE, one cycle: simulator.rs:529-813
    {
        ProcessingElement &pe = pes[p];
        if (a_pending[p] > 0) {
            --a_pending[p];
            return;
        }
        const uint64_t a0 = a.read_words, b0 = b.read_words, pr0 = psum.read_words, pw0 = psum.write_words;
        if (pe.has_task && pe.drain_since == NONE && Planner::window_done(*pe.win)) pe.drain_since = cycle;
        if ((!pe.has_task || Planner::window_done(*pe.win)) && pe.idle()) {
            if (pe.mem_finish == NONE) {
                if (pe.has_task) {
                    const uint64_t mem_cycles = (uint64_t)((float)pe.task.traffic / pe_bandwidth());
                    if (!pe.task.merge)   // latency of the block, for the height policy (:581-596)
                        plan.blocks[pe.task.block].latency += std::max(cycle - pe.task.start_cycle, mem_cycles);
                    pe.mem_finish = pe.task.start_cycle + mem_cycles;
                    // cycles the datapath spent draining after the memory side had finished are discounted (:608-624)
                    const uint64_t drain = pe.drain_since != NONE ? cycle - pe.drain_since : 0;
                    if (cycle > pe.mem_finish && pe.config_unchanged) drain_cycles[p] += cycle - std::max(cycle - drain, pe.mem_finish);
                }
            }
            if (pe.mem_finish != NONE && pe.mem_finish > cycle) return;   // the memory side is still busy
            if (pe.has_task) plan.retire_window(pe.task.window);
            if (pe.has_task && !pe.task.merge && plan.block_done(pe.task.block)) {
                plan.mark_tail_rows(pe.task.block);
                if (cfg.accelerator == SPADA_ACCEL_SPADA) {   // update_group_cost, rowwise_perf_adjust.rs:233-248
                    const Block &k = plan.blocks[pe.task.block];
                    uint64_t elems = 0;
                    for (uint64_t t : k.total) elems += t;
                    plan.policy.record(k.anchor[0], k.shape[0], k.latency, elems);
                }
            }
            settle_finished_rows();
            Task t;
            uint64_t latency = 0;
            const bool got = plan.assign(pe.has_task, pe.has_task ? pe.task.block : 0, a, cycle, &latency, &t);
            a_pending[p] += pe.set_task(got, latency, std::move(t));
            pe.win = pe.has_task ? &plan.windows.at(pe.task.window) : nullptr;
        }
        if (!pe.has_task) return;

        // fetch: refill every stream buffer to sb_size elements (:686-696)
        for (uint64_t l = 0; l < pe.lanes; ++l) {
            uint64_t held = 0;
            for (const Elem &e : pe.stream[l]) held += e.row != NONE;
            const bool more = stream_lane(pe, l, pe.sb_size - std::min(held, pe.sb_size), scratch);
            pe.feed(l, more, scratch);
        }
        // multiply: last cycle's products move to the psum buffers, new operands are taken (:698-734)
        std::vector<Slot> &bs = pe.bs_buf;
        bs.assign(pe.lanes, Slot{});
        for (uint64_t l = 0; l < pe.lanes; ++l) {
            pe.full[l] = pe.psum[l].size() >= pe.pb_size - 1;
            bs[l] = pe.a[l].has ? pe.take_b(l) : Slot{};
        }
        std::vector<Slot> &prods = pe.prods_buf;
        prods = pe.prod;
        pe.multiply(bs);
        for (uint64_t l = 0; l < pe.lanes; ++l)
            if (prods[l].has) pe.psum[l].push_back(prods[l].e);
        // collect: up to two products per lane below the group's tail flag enter the sorting network (:736-755)
        pe.update_tails();
        {
            Staged batch;
            if (!pe.spare.empty()) {
                batch = std::move(pe.spare.back());
                pe.spare.pop_back();
            }
            batch.wait = pe.sn_latency;
            const uint64_t g = pe.group_lanes ? pe.group_lanes : 1;
            batch.groups.resize((pe.lanes + g - 1) / g);
            for (auto &v : batch.groups) v.clear();
            bool any = false;
            for (uint64_t l = 0; l < pe.lanes; ++l)
                for (uint64_t k = 0; k < pe.pops; ++k)
                    if (!pe.psum[l].empty() && pe.psum[l].front().col < pe.tail[l]) {
                        batch.groups[l / g].push_back(pe.psum[l].front());
                        pe.psum[l].pop_front();
                        any = true;
                    }
            if (any) pe.sorter.push_back(std::move(batch));
            else pe.spare.push_back(std::move(batch));
        }
        // sort, then add runs of equal column (:757-763; SortingNetwork :143-171, MergeTree :199-230)
        Staged sorted;
        if (ProcessingElement::release(pe.sorter, &sorted)) {
            for (auto &g : sorted.groups)
                std::stable_sort(g.begin(), g.end(), [](const Elem &x, const Elem &y) { return x.col < y.col; });
            sorted.wait = pe.mt_latency;
            pe.merger.push_back(std::move(sorted));
        }
        Staged merged;
        if (ProcessingElement::release(pe.merger, &merged)) {
            for (auto &g : merged.groups) {   // runs of equal column are added left to right, in place (simulator.rs:213-218)
                size_t w = 0;
                for (size_t r = 0; r < g.size(); ++r) {
                    if (w && g[w - 1].col == g[r].col) g[w - 1].val += g[r].val;
                    else g[w++] = g[r];
                }
                g.resize(w);
            }
            write_psums(pe, merged);
            pe.spare.push_back(std::move(merged));
        }
        // memory words this PE moved in this cycle count towards its task (:790-800)
        if (pe.has_task)
            pe.task.traffic += (a.read_words - a0) + (b.read_words - b0) + (psum.read_words - pr0) + (psum.write_words - pw0);
    }

    void tree_cycle(uint64_t i)   // adder_tree_exec, simulator.rs:1093-1181
    {
        TreeMerger &t = trees[i];
        if ((!t.has_task || Planner::window_done(*t.win)) && t.idle()) {
            if (t.has_task) plan.retire_window(t.task.window);
            settle_finished_rows();
            Task nt;
            bool got = false;
            if (t.failed_at[0] != plan.fibers_changed || t.failed_at[1] != cache.psum_membership) {   // (else: nothing it looks at changed)
                got = plan.tree_merge_task(t.width, cache, cycle, &nt);
                t.failed_at[0] = got ? NONE : plan.fibers_changed;
                t.failed_at[1] = got ? NONE : cache.psum_membership;
            }
            t.set_task(got, std::move(nt));
            t.win = t.has_task ? &plan.windows.at(t.task.window) : nullptr;
        }
        if (!t.has_task) return;
        Window &w = *t.win;
        for (uint64_t l = 0; l < t.width; ++l) {
            if (t.node[t.depth][l].has) continue;
            // stream_b_element, :1198-1250: one element of the lane's fiber, or the end marker
            Elem e = END_MARK;
            const auto src = w.lane_src[l];
            if (src.second != NONE && w.streamed[l] < w.lane_len[l]) {   // (an exhausted fiber is gone from cache and DRAM: asking
                                                                         // again, as the reference does every cycle, finds nothing)
                cache.consume(src.second, w.streamed[l], 1, scratch);
                if (!scratch.empty()) {
                    ++w.streamed[l];
                    e = scratch.back();
                }
            }
            t.push_leaf(l, e);
        }
        const Slot top = t.step();
        const Slot product = t.c;
        t.b = (top.has && !is_end(top.e)) ? top : Slot{};
        t.multiply();
        const Slot out = t.add(product);
        if (out.has) {   // adder_tree_write_psum, :1183-1196
            const uint64_t fiber = w.outputs[0].second;
            ++plan.psum_len[fiber];
            cache.append_element(fiber, Elem{fiber, out.e.col, out.e.val});
        }
    }

    bool all_quiet() const   // :815-824
    {
        if (!plan.traversed) return false;
        for (const auto &pe : pes)
            if (!pe.idle() || pe.has_task) return false;
        for (const auto &t : trees)
            if (!t.idle() || t.has_task) return false;
        return true;
    }

    void assemble()   // get_exec_result, simulator.rs:1034-1062
    {
        const uint64_t n = a.rows();
        std::vector<const Fiber *> by_raw(n, nullptr);
        for (uint64_t r = 0; r < n; ++r) {
            if (a.row_len(r) == 0) continue;
            auto f = plan.finished.find(r);
            if (f == plan.finished.end()) continue;   // (a row never marked finished comes out empty: :1044)
            const Fiber *fib = nullptr;
            auto d = psum.data.find(f->second);
            if (d != psum.data.end()) fib = &d->second;
            else {
                auto c = cache.rows.find(f->second);
                if (c == cache.rows.end()) panic("result: the final fiber of a row is neither in the psum DRAM nor in the cache");
                fib = &c->second;
            }
            by_raw[a.raw(r)] = fib;
        }
        c_indptr.assign(n + 1, 0);
        for (uint64_t r = 0; r < n; ++r) c_indptr[r + 1] = c_indptr[r] + (by_raw[r] ? by_raw[r]->len() : 0);
        c_indices.resize(c_indptr[n]);
        c_data.resize(c_indptr[n]);
        for (uint64_t r = 0; r < n; ++r)
            if (by_raw[r]) {
                std::copy(by_raw[r]->cols.begin(), by_raw[r]->cols.end(), c_indices.begin() + c_indptr[r]);
                std::copy(by_raw[r]->vals.begin(), by_raw[r]->vals.end(), c_data.begin() + c_indptr[r]);
            }
    }
};

// ---- C entry points ----------------------------------------------------------------------------------------------------
extern "C" int spada_cycle_create(const spada_cycle_config *cfg, const spada_csr_view *A, const spada_csr_view *B,
                                  const uint64_t *row_remap, spada_cycle_model **out)
{
    spada::clear_error();
    if (!cfg || !A || !B || !out) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: null argument");
    if (cfg->struct_size != sizeof(spada_cycle_config)) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: struct_size mismatch");
    if (A->cols != B->rows) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: A has %llu columns, B has %llu rows",
                                               (unsigned long long)A->cols, (unsigned long long)B->rows);
    if (cfg->pe_num == 0 || cfg->lane_num == 0 || cfg->word_byte == 0 || cfg->freq <= 0.0f || cfg->channel == 0 ||
        cfg->bandwidth_per_channel <= 0.0f || cfg->block_shape[0] == 0 || cfg->block_shape[1] == 0)
        return spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: pe_num, lane_num, word_byte, freq, channel, bandwidth and block_shape must be positive");
    if ((cfg->lane_num & (cfg->lane_num - 1)) != 0 || cfg->lane_num < 2)
        return spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: lane_num must be a power of two >= 2");
    if (cfg->accelerator < SPADA_ACCEL_IP || cfg->accelerator > SPADA_ACCEL_SPADA)
        return spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: unknown accelerator %d", cfg->accelerator);
    try {
        auto *m = new spada_cycle_model();
        m->cfg = *cfg;
        m->a.m = *A;
        m->a.remap = row_remap;
        m->b.m = *B;
        m->cache.capacity = cfg->cache_size / cfg->word_byte;
        m->cache.psum_base = B->rows + 1;   // output_base_addr = dram_b.indptr.len(), main.rs:65
        m->cache.b = &m->b;
        m->cache.psum = &m->psum;
        m->words_per_cycle_channel = cfg->bandwidth_per_channel / cfg->freq / (float)cfg->word_byte;
        Planner &p = m->plan;
        p.lanes = cfg->lane_num;
        p.n_rows = A->rows;
        p.psum_base = m->cache.psum_base;
        p.next_addr = p.psum_base;
        p.accel = cfg->accelerator;
        p.cache_latency = cfg->cache_latency;
        // default block shape per accelerator, main.rs:67-72
        p.shape[0] = cfg->accelerator == SPADA_ACCEL_OP ? cfg->lane_num : cfg->block_shape[0];
        p.shape[1] = cfg->accelerator == SPADA_ACCEL_OP ? 1 : cfg->block_shape[1];
        if (p.shape[0] > cfg->lane_num || cfg->lane_num % p.shape[0] != 0)
            return delete m, spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: block_shape[0] must divide lane_num");
        p.a_len.resize(A->rows);
        for (uint64_t r = 0; r < A->rows; ++r) p.a_len[r] = m->a.row_len(r);
        p.a_assigned.assign(A->rows, 0);
        p.b_len.resize(B->rows);
        for (uint64_t r = 0; r < B->rows; ++r) p.b_len[r] = B->indptr[r + 1] - B->indptr[r];
        p.policy.lanes = cfg->lane_num;
        if (A->rows) p.policy.parse(m->a, 1.5f);   // var_factor, simulator.rs:449
        m->pes.resize(cfg->pe_num);
        for (auto &pe : m->pes) pe.init(cfg->lane_num);   // sb 4, pb 8, 2 pops, latencies 4 / 4: simulator.rs:450-454
        m->trees.resize(cfg->at_num);
        for (auto &t : m->trees) t.init(8);               // tree_width, simulator.rs:455
        m->a_pending.assign(cfg->pe_num, 0);
        m->drain_cycles.assign(cfg->pe_num, 0);
        *out = m;
        return SPADA_OK;
    } catch (const std::exception &e) {
        return spada::fail(SPADA_ERR_INVALID, "spada_cycle_create: %s", e.what());
    }
}

extern "C" int spada_cycle_execute(spada_cycle_model *m, uint64_t max_cycles)
{
    spada::clear_error();
    if (!m) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_execute: null model");
    if (m->executed) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_execute: the model has been executed already");
    try {
        m->cycle = 0;
        for (;;) {   // Simulator::execute, simulator.rs:509-890
            for (uint64_t p = 0; p < m->pes.size(); ++p) m->pe_cycle(p);
            for (uint64_t t = 0; t < m->trees.size(); ++t) m->tree_cycle(t);
            if (m->all_quiet()) break;
            ++m->cycle;
            if (max_cycles && m->cycle > max_cycles)
                return spada::fail(SPADA_ERR_UNSUPPORTED, "spada_cycle_execute: more than %llu cycles", (unsigned long long)max_cycles);
        }
        m->assemble();
        m->executed = true;
        return SPADA_OK;
    } catch (const std::exception &e) {
        return spada::fail(SPADA_ERR_INVALID, "spada_cycle_execute: cycle %llu: %s", (unsigned long long)m->cycle, e.what());
    }
}

extern "C" int spada_cycle_get_counts(const spada_cycle_model *m, spada_cycle_counts *out)
{
    spada::clear_error();
    if (!m || !out) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_get_counts: null argument");
    if (out->struct_size != sizeof(spada_cycle_counts)) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_get_counts: struct_size mismatch");
    if (!m->executed) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_get_counts: execute first");
    const uint64_t discount = m->drain_cycles.empty() ? 0 : *std::min_element(m->drain_cycles.begin(), m->drain_cycles.end());
    out->exec_cycles = m->cycle - discount;   // get_exec_cycle, simulator.rs:1030-1032
    out->raw_cycles = m->cycle;
    out->a_read = m->a.read_words;
    out->a_write = m->a.write_words;
    out->b_read = m->b.read_words;
    out->b_write = m->b.write_words;
    out->c_read = m->psum.read_words;
    out->c_write = m->psum.write_words;
    out->cache_read = m->cache.read_words;
    out->cache_write = m->cache.write_words;
    out->cache_miss = m->cache.miss_words;
    out->b_evict = m->cache.b_evict_words;
    out->psum_evict = m->cache.psum_evict_words;
    out->blocks = m->plan.blocks.size() - m->plan.n_pe_merges - m->plan.n_tree_merges;
    out->windows = m->plan.n_windows;
    out->pe_merge_tasks = m->plan.n_pe_merges;
    out->tree_merge_tasks = m->plan.n_tree_merges;
    out->c_nnz = m->c_indptr.empty() ? 0 : m->c_indptr.back();
    return SPADA_OK;
}

extern "C" int spada_cycle_get_result(const spada_cycle_model *m, uint64_t *c_indptr, uint64_t *c_indices, double *c_data)
{
    spada::clear_error();
    if (!m || !c_indptr) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_get_result: null argument");
    if (!m->executed) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_get_result: execute first");
    std::copy(m->c_indptr.begin(), m->c_indptr.end(), c_indptr);
    if (!m->c_indices.empty()) {
        if (!c_indices || !c_data) return spada::fail(SPADA_ERR_INVALID, "spada_cycle_get_result: null output array");
        std::copy(m->c_indices.begin(), m->c_indices.end(), c_indices);
        std::copy(m->c_data.begin(), m->c_data.end(), c_data);
    }
    return SPADA_OK;
}

extern "C" void spada_cycle_destroy(spada_cycle_model *m) { delete m; }
