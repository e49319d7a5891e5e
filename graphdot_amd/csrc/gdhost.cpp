// libgdhost.so -- native host side of the MI355X marginalized-graph-kernel
// path (include/gdhost.h): graph packer, label-class numbering, solver-variant
// classification and job layout.  Host code only (g++), no HIP.
//
// The numpy implementations (_devicegraph.pack_many / _label_classes,
// HIPBackend._classify_pairs / _partition) are the specification; every
// function here reproduces their results byte for byte (tests/test_host_model
// .py, tests/test_abi_and_host_logic.py).
#include "gdhost.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr int64_t ALIGN = 16;
inline int64_t pad(int64_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

// (a, b) for a, b in [0, 15], sorted by descending a * b, ties row-major:
// the order of the degree-pair rectangles in the sorted row space
// (mgk_oc.h class_order; zero-size rectangles of a smaller degree bound do
// not change the relative order of the others)
struct rect_order_t {
    int a[256], b[256], prod[257];
    // the rectangles of classes <= d, in order, for every degree bound d
    int n_upto[16], upto[16][256];
    rect_order_t() {
        int n = 0;
        for (int p = 15 * 15; p >= 0; --p)
            for (int x = 0; x <= 15; ++x)
                for (int y = 0; y <= 15; ++y)
                    if (x * y == p) {
                        a[n] = x;
                        b[n] = y;
                        prod[n] = p;
                        ++n;
                    }
        prod[256] = 0;
        for (int d = 0; d < 16; ++d) {
            n_upto[d] = 0;
            for (int k = 0; k < 256; ++k)
                if (a[k] <= d && b[k] <= d) upto[d][n_upto[d]++] = k;
        }
    }
};
const rect_order_t RECT;

// Worker threads of the passes over job lists and class pairs: GD_HOST_THREADS,
// default min(8, hardware threads); lists below `min_items` per thread stay
// on the calling thread.
inline int host_threads(int64_t n, int64_t min_items) {
    static const int configured = [] {
        const char *e = std::getenv("GD_HOST_THREADS");
        int t = e ? std::atoi(e) : 0;
        if (t <= 0) {
            t = (int)std::thread::hardware_concurrency();
            t = t <= 0 ? 1 : (t > 8 ? 8 : t);
        }
        return t > 64 ? 64 : t;
    }();
    const int64_t by_size = min_items > 0 ? n / min_items : n;
    return (int)std::max<int64_t>(1, std::min<int64_t>(configured, by_size));
}

// fn(begin, end, thread index) over [0, n) cut into T contiguous ranges.
// Every range runs exactly once: a range whose thread cannot be created
// (std::system_error under a pid / thread limit -- containers, eight ranks per
// node) runs on the calling thread; an exception inside a worker is carried to
// the caller (a std::thread that lets one escape calls std::terminate) and
// re-thrown there, where the entry points' function-try-blocks turn it into
// an error code.
template<class F> inline void parallel_ranges(int64_t n, int T, F &&fn) {
    if (T <= 1) {
        fn((int64_t)0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    std::vector<char> started((size_t)T, 0);
    std::exception_ptr failure;
    std::mutex failure_lock;
    auto guarded = [&](int k) {
        try {
            fn(n * k / T, n * (k + 1) / T, k);
        } catch (...) {
            std::lock_guard<std::mutex> hold(failure_lock);
            if (!failure) failure = std::current_exception();
        }
    };
    try {
        pool.reserve((size_t)T - 1);
        for (int k = 1; k < T; ++k) {
            pool.emplace_back(guarded, k);
            started[(size_t)k] = 1;
        }
    } catch (...) {
        // no more threads to be had: the remaining ranges run here
    }
    guarded(0);
    for (int k = 1; k < T; ++k)
        if (!started[(size_t)k]) guarded(k);
    for (auto &th : pool) th.join();
    if (failure) std::rethrow_exception(failure);
}

}  // namespace

extern "C" {

const char *gdh_version(void) { return "gdhost 5.0 (round 5)"; }

int gdh_pack_graphs(int32_t G, const int64_t *node_off, const int64_t *edge_off,
                    const int64_t *node_id, const int64_t *ei, const int64_t *ej,
                    const float *w, const uint8_t *node_rec, int32_t node_size,
                    const uint8_t *label_rec, int32_t label_size,
                    int32_t edge_size, int32_t label_offset, int32_t weight_bytes,
                    uint8_t *blob, int64_t blob_capacity, int64_t *blob_off,
                    int64_t *sec_off, int64_t *nnz, uint16_t *perm, int64_t *rank,
                    float *degree, int64_t *count, uint16_t *rowptr, uint16_t *nz,
                    int64_t *eid, int64_t nz_capacity, int64_t *nz_off,
                    int64_t *maxdeg) try {
    if (G < 0 || node_size < 0 || label_size < 0 || edge_size < 0) return -1;
    if (weight_bytes != 0 && weight_bytes != 4 && weight_bytes != 8) return -1;
    struct entry_t {
        int64_t key;
        int32_t src, dst, e;
    };
    // Three passes: (A) per graph, in parallel -- degrees, directed nonzeros,
    // the degree renumbering and everything addressed by the graph's node /
    // edge offsets; the nonzeros wait in a scratch array at twice the edge
    // offset (their upper bound); (B) serial -- the running offsets of the
    // nonzero arrays and of the blobs, which need every graph's nonzero count;
    // (C) per graph, in parallel -- nz / eid and the blobs.
    if (G == 0) {
        blob_off[0] = 0;
        nz_off[0] = 0;
        return 0;
    }
    const int64_t total_m = edge_off[G] - edge_off[0];
    if (total_m < 0) return -1;
    std::vector<entry_t> all_uniq((size_t)(2 * total_m));
    std::atomic<int> status{0};
    const int T = host_threads(G, 64);
    parallel_ranges(G, T, [&](int64_t g0, int64_t g1, int) {
        std::vector<entry_t> ent;
        std::vector<float> deg;
        std::vector<int64_t> cnt;
        std::vector<int32_t> order;
        for (int64_t g = g0; g < g1 && status.load(std::memory_order_relaxed) == 0; ++g) {
            const int64_t n0 = node_off[g], n = node_off[g + 1] - n0;
            const int64_t e0 = edge_off[g], m = edge_off[g + 1] - e0;
            if (n < 0 || m < 0 || n > 0xFFFF || 2 * m > 0xFFFF) {
                status = -1;
                return;
            }
            // ---- degrees: float32 sums in the order of the per-graph packer
            // (ei pass, ej pass, self loops taken back once) ----------------
            deg.assign((size_t)n, 0.f);
            for (int64_t e = 0; e < m; ++e) {
                const int64_t a = ei[e0 + e], b = ej[e0 + e];
                if (a < 0 || a >= n || b < 0 || b >= n) {
                    status = -1;
                    return;
                }
            }
            for (int64_t e = 0; e < m; ++e) deg[(size_t)ei[e0 + e]] += w ? w[e0 + e] : 1.f;
            for (int64_t e = 0; e < m; ++e) deg[(size_t)ej[e0 + e]] += w ? w[e0 + e] : 1.f;
            for (int64_t e = 0; e < m; ++e)
                if (ei[e0 + e] == ej[e0 + e]) deg[(size_t)ei[e0 + e]] -= w ? w[e0 + e] : 1.f;
            for (int64_t i = 0; i < n; ++i)
                if (deg[(size_t)i] == 0.f) deg[(size_t)i] = 1.f;
            // ---- directed nonzeros: both orientations, duplicates collapse
            // onto their first occurrence (forward orientations first) -------
            ent.resize((size_t)(2 * m));
            for (int64_t e = 0; e < m; ++e) {
                const int32_t a = (int32_t)ei[e0 + e], b = (int32_t)ej[e0 + e];
                ent[(size_t)e] = {a * n + b, a, b, (int32_t)e};
                ent[(size_t)(m + e)] = {b * n + a, b, a, (int32_t)e};
            }
            std::stable_sort(ent.begin(), ent.end(),
                             [](entry_t const &x, entry_t const &y) { return x.key < y.key; });
            entry_t *const uniq = all_uniq.data() + 2 * (e0 - edge_off[0]);
            int64_t z = 0;
            for (size_t k = 0; k < ent.size(); ++k)
                if (k == 0 || ent[k].key != ent[k - 1].key) uniq[z++] = ent[k];
            // ---- renumber the nodes by descending adjacency count (stable) -
            cnt.assign((size_t)n, 0);
            for (int64_t k = 0; k < z; ++k) ++cnt[(size_t)uniq[k].src];
            order.resize((size_t)n);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(),
                             [&](int32_t x, int32_t y) { return cnt[(size_t)x] > cnt[(size_t)y]; });
            int64_t md = 0;
            for (int64_t k = 0; k < n; ++k) {
                const int32_t old = order[(size_t)k];
                perm[n0 + k] = (uint16_t)old;
                rank[n0 + old] = k;
                degree[n0 + k] = deg[(size_t)old];
                count[n0 + k] = cnt[(size_t)old];
                md = std::max(md, cnt[(size_t)old]);
            }
            maxdeg[g] = md;
            for (int64_t k = 0; k < z; ++k) {
                entry_t &u = uniq[k];
                u.src = (int32_t)rank[n0 + u.src];
                u.dst = (int32_t)rank[n0 + u.dst];
                u.key = (int64_t)u.src * n + u.dst;
            }
            std::sort(uniq, uniq + z,
                      [](entry_t const &x, entry_t const &y) { return x.key < y.key; });
            uint16_t *rp = rowptr + n0 + g;
            rp[0] = 0;
            for (int64_t k = 0; k < n; ++k) rp[k + 1] = (uint16_t)(rp[k] + count[n0 + k]);
            nnz[g] = z;
            for (int64_t r = 0; r < n; ++r) {
                const int64_t id = node_id[n0 + r];
                if (id < 0 || id >= n) {
                    status = -1;
                    return;
                }
            }
        }
    });
    if (status != 0) return status;
    // ---- (B) offsets -------------------------------------------------------
    int64_t cursor = 0, zc = 0;
    blob_off[0] = 0;
    nz_off[0] = 0;
    for (int32_t g = 0; g < G; ++g) {
        const int64_t n = node_off[g + 1] - node_off[g], z = nnz[g];
        const int64_t sizes[6] = {4 * n, (int64_t)node_size * n, 2 * (n + 1), 4 * z,
                                  (int64_t)edge_size * z, 2 * n};
        int64_t c = 0;
        for (int s_ = 0; s_ < 6; ++s_) {
            sec_off[(size_t)g * 6 + s_] = c;
            c += pad(sizes[s_]);
        }
        cursor += std::max<int64_t>(c, ALIGN);
        zc += z;
        if (zc > nz_capacity || cursor > blob_capacity) return -2;
        blob_off[g + 1] = cursor;
        nz_off[g + 1] = zc;
    }
    // ---- (C) flat nonzero views and blobs ------------------------------------
    parallel_ranges(G, T, [&](int64_t g0, int64_t g1, int) {
        for (int64_t g = g0; g < g1; ++g) {
            const int64_t n0 = node_off[g], n = node_off[g + 1] - n0;
            const int64_t e0 = edge_off[g], z = nnz[g], z0 = nz_off[g];
            entry_t const *const uniq = all_uniq.data() + 2 * (e0 - edge_off[0]);
            for (int64_t k = 0; k < z; ++k) {
                nz[2 * (z0 + k)] = (uint16_t)uniq[k].src;
                nz[2 * (z0 + k) + 1] = (uint16_t)uniq[k].dst;
                eid[z0 + k] = uniq[k].e;
            }
            const int64_t *off = sec_off + (size_t)g * 6;
            uint8_t *B = blob + blob_off[g];
            std::memset(B, 0, (size_t)(blob_off[g + 1] - blob_off[g]));
            std::memcpy(B + off[0], degree + n0, (size_t)(4 * n));
            for (int64_t r = 0; r < n; ++r)      // node row r -> its new index
                std::memcpy(B + off[1] + rank[n0 + node_id[n0 + r]] * node_size,
                            node_rec + (size_t)(n0 + r) * node_size, (size_t)node_size);
            std::memcpy(B + off[2], rowptr + n0 + g, (size_t)(2 * (n + 1)));
            std::memcpy(B + off[3], nz + 2 * z0, (size_t)(4 * z));
            for (int64_t k = 0; k < z; ++k) {
                uint8_t *rec = B + off[4] + k * edge_size;
                const int64_t e = e0 + eid[z0 + k];
                if (weight_bytes == 4) {
                    const float v = w ? w[e] : 1.f;
                    std::memcpy(rec, &v, 4);
                } else if (weight_bytes == 8) {
                    const double v = (double)(w ? w[e] : 1.f);
                    std::memcpy(rec, &v, 8);
                }
                if (label_size > 0)
                    std::memcpy(rec + label_offset, label_rec + (size_t)e * label_size,
                                (size_t)label_size);
            }
            std::memcpy(B + off[5], perm + n0, (size_t)(2 * n));
        }
    });
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

int gdh_number_records(const uint8_t *rec, int64_t n, int32_t itemsize,
                       const int32_t *part_off, const int32_t *part_len,
                       int32_t n_parts, int32_t *cls, int64_t *first,
                       int64_t *n_classes) try {
    if (n < 0 || itemsize < 0 || n_parts < 0) return -1;
    int64_t klen = 0;
    for (int32_t p = 0; p < n_parts; ++p) {
        if (part_off[p] < 0 || part_len[p] < 0 || part_off[p] + part_len[p] > itemsize) return -1;
        klen += part_len[p];
    }
    *n_classes = 0;
    if (n == 0) return 0;
    if (klen == 0) {      // no key bytes: one class
        for (int64_t i = 0; i < n; ++i) cls[i] = 0;
        first[0] = 0;
        *n_classes = 1;
        return 0;
    }
    const int64_t stride = klen <= 8 ? 8 : klen;
    std::vector<uint8_t> keys((size_t)(n * stride), 0);
    for (int64_t i = 0; i < n; ++i) {
        uint8_t *k = keys.data() + i * stride;
        for (int32_t p = 0; p < n_parts; ++p) {
            std::memcpy(k, rec + i * itemsize + part_off[p], (size_t)part_len[p]);
            k += part_len[p];
        }
    }
    // Distinct keys through a hash table (first occurrence kept: what a
    // stable sort of all records would put first), then only THEY are sorted
    // into np.unique's order -- a data set has a few dozen label classes among
    // tens of thousands of records (a stable sort of every record took 1 ms
    // of a 2.4 ms arena on the GPU box's host).
    auto row = [&](int64_t i) { return keys.data() + i * stride; };
    auto hash = [&](const uint8_t *k) {
        uint64_t h = 1469598103934665603ull;
        if (klen <= 8) {
            uint64_t v;
            std::memcpy(&v, k, 8);
            h = (v ^ (v >> 29)) * 0x9E3779B97F4A7C15ull;
            return h ^ (h >> 32);
        }
        for (int64_t t = 0; t < klen; ++t) h = (h ^ k[t]) * 1099511628211ull;
        return h ^ (h >> 32);
    };
    size_t cap = 64;
    while (cap < (size_t)n * 2) cap <<= 1;
    std::vector<int64_t> slot(cap, -1);          // index into `distinct`
    std::vector<int64_t> distinct;               // first record of every distinct key
    std::vector<int32_t> tmp((size_t)n);         // record -> position in `distinct`
    for (int64_t i = 0; i < n; ++i) {
        size_t at = (size_t)hash(row(i)) & (cap - 1);
        for (;;) {
            const int64_t d = slot[at];
            if (d < 0) {
                slot[at] = (int64_t)distinct.size();
                tmp[(size_t)i] = (int32_t)distinct.size();
                distinct.push_back(i);
                break;
            }
            if (std::memcmp(row(distinct[(size_t)d]), row(i), (size_t)stride) == 0) {
                tmp[(size_t)i] = (int32_t)d;
                break;
            }
            at = (at + 1) & (cap - 1);
        }
    }
    const int64_t nc = (int64_t)distinct.size();
    std::vector<int64_t> ord((size_t)nc);
    std::iota(ord.begin(), ord.end(), 0);
    if (klen <= 8) {      // little-endian unsigned integers
        const uint64_t *k64 = reinterpret_cast<const uint64_t *>(keys.data());
        std::sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) {
            return k64[distinct[(size_t)x]] < k64[distinct[(size_t)y]];
        });
    } else {              // byte rows, lexicographic
        std::sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) {
            return std::memcmp(row(distinct[(size_t)x]), row(distinct[(size_t)y]), (size_t)klen) < 0;
        });
    }
    std::vector<int32_t> number((size_t)nc);
    for (int64_t c = 0; c < nc; ++c) {
        number[(size_t)ord[(size_t)c]] = (int32_t)c;
        first[c] = distinct[(size_t)ord[(size_t)c]];
    }
    for (int64_t i = 0; i < n; ++i) cls[i] = number[(size_t)tmp[(size_t)i]];
    *n_classes = nc;
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

int gdh_classify_oc(int64_t n_pairs, const int32_t *ca, const int32_t *cb,
                    const int32_t *n_node, const int32_t *n_nz,
                    const int64_t *image_bytes, const int32_t *maxdeg,
                    const uint16_t *hist, int32_t n_var, const int32_t *W,
                    const int32_t *S, const int32_t *R, const int32_t *D,
                    const int32_t *n_L, const int32_t *L, int32_t C,
                    int32_t real_size, int64_t lds_limit, int32_t fly_min_degree,
                    const int64_t *extra_lds, int32_t *choice, int64_t *NP) try {
    if (n_pairs < 0 || n_var < 0 || (C != 1 && C != 2)) return -1;
    (void)n_nz;
    constexpr int MAXL = 12;
    parallel_ranges(n_pairs, host_threads(n_pairs, 2048), [&](int64_t t0, int64_t t1, int) {
    for (int64_t t = t0; t < t1; ++t) {
        const int32_t a = ca[t], b = cb[t];
        const int64_t n1 = n_node[a], n2 = n_node[b];
        const int64_t N = n1 * n2, np_ = n1 * (n2 | 1);
        NP[t] = np_;
        choice[t] = -1;
        const int pmd_true = std::max(maxdeg[a], maxdeg[b]);
        // (the histograms resolve degrees up to 14: graphs beyond that can
        // only take the on-the-fly variants, S = 0)
        const int pmd = std::min(pmd_true, 15);
        const int64_t gbytes = std::max(image_bytes[a], image_bytes[b]);
        // cumulative sizes of the degree-pair rectangles in sorted row order
        // (only the rectangles of classes <= pmd can be non-empty)
        int64_t cum[256];
        int prods[256], nr = 0;
        {
            const uint16_t *h1 = hist + (size_t)a * 16, *h2 = hist + (size_t)b * 16;
            int64_t c = 0;
            for (int u = 0; u < RECT.n_upto[pmd]; ++u) {
                const int k = RECT.upto[pmd][u];
                const int64_t sz = (int64_t)h1[RECT.a[k]] * h2[RECT.b[k]];
                if (sz == 0) continue;
                c += sz;
                cum[nr] = c;
                prods[nr] = RECT.prod[k];
                ++nr;
            }
        }
        auto trip_at = [&](int64_t first) -> int {   // product of sorted row `first`
            if (first >= N) return 0;
            int k = 0;
            while (k < nr && cum[k] <= first) ++k;
            return k < nr ? prods[k] : 0;
        };
        // one-wave walk (rows 0, 64, 128, ...): the trips of the first
        // batches in one merged pass, shared by every W = 1 variant
        int trips1[MAXL + 1];
        {
            int k = 0;
            for (int bt = 0; bt <= MAXL; ++bt) {
                const int64_t first = 64 * (int64_t)bt;
                while (k < nr && cum[k] <= first) ++k;
                trips1[bt] = (first < N && k < nr) ? prods[k] : 0;
            }
        }
        for (int32_t v = 0; v < n_var; ++v) {
            const bool fly = S[v] == 0;
            if (!fly && (pmd_true > D[v] || pmd_true > 14)) continue;
            const int64_t T = 64 * (int64_t)W[v];
            if (N > T * R[v] || np_ >= (fly ? 0x3FFF : 0xFFFF)) continue;
            const int64_t NR = T * R[v];
            const int64_t pcap = (np_ + 1 + 3) / 4 * 4;
            const int64_t NRy = ((n_L[v] > 0 && C != 2) || fly) ? 0 : NR;
            const int64_t lds = (pcap + NRy) * C * real_size + 4 * NR + 2 * gbytes +
                                4 * (int64_t)W[v] * real_size + 4 * (D[v] > 6 ? 128 : 64) + 256 + 16 +
                                (extra_lds ? extra_lds[v] : 0);
            if (lds > lds_limit) continue;
            const int64_t nb = (N + T - 1) / T;
            bool ok = true;
            if (fly) {             // on-the-fly: the high-degree pairs
                ok = pmd_true > fly_min_degree;
            } else if (n_L[v] > 0) {      // static layout: every batch under its segment
                for (int64_t k = 0; k < nb && ok; ++k) {
                    const int cap = k < n_L[v] && k < MAXL ? L[(size_t)v * MAXL + k] : 0;
                    ok = (W[v] == 1 && k <= MAXL ? trips1[k] : trip_at(k * T)) <= cap;
                }
            } else if (W[v] == 1 && nb <= MAXL + 1) {
                int64_t total = 0;
                for (int64_t k = 0; k < nb; ++k) total += trips1[k];
                ok = total <= S[v];
            } else {               // dynamic: the heaviest wave's slot total
                int64_t worst = 0;
                for (int32_t wv = 0; wv < W[v]; ++wv) {
                    int64_t total = 0;
                    for (int64_t k = 0; k < nb; ++k)
                        total += trip_at(k * T + 64 * ((k & 1) ? W[v] - 1 - wv : wv));
                    worst = std::max(worst, total);
                }
                ok = worst <= S[v];
            }
            if (ok) {
                choice[t] = v;
                break;
            }
        }
    }
    });
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

int gdh_pair_keys(const uint32_t *jobs, int64_t n_jobs, const int32_t *cid,
                  int32_t n_graphs, int32_t nc, int32_t *pk, int64_t *count) try {
    if (n_jobs < 0 || nc <= 0) return -1;
    const size_t nk = (size_t)nc * (size_t)nc;
    const int T = host_threads(n_jobs, 65536);
    std::vector<std::vector<int64_t>> local((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    parallel_ranges(n_jobs, T, [&](int64_t t0, int64_t t1, int k) {
        std::vector<int64_t> &cnt = local[(size_t)k];
        cnt.assign(nk, 0);
        for (int64_t t = t0; t < t1; ++t) {
            const uint32_t i = jobs[2 * t], j = jobs[2 * t + 1];
            if (i >= (uint32_t)n_graphs || j >= (uint32_t)n_graphs) {
                bad[(size_t)k] = 1;
                return;
            }
            const int32_t key = cid[i] * nc + cid[j];
            pk[t] = key;
            ++cnt[(size_t)key];
        }
    });
    for (int k = 0; k < T; ++k)
        if (bad[(size_t)k]) return -1;
    for (size_t key = 0; key < nk; ++key) {
        int64_t c = 0;
        for (int k = 0; k < T; ++k) c += local[(size_t)k][key];
        count[key] = c;
    }
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

int gdh_order_jobs(const int32_t *pk, int64_t n_jobs, const int32_t *rank_of_key,
                   int64_t n_keys, int64_t n_ranks, uint32_t *order,
                   const uint32_t *jobs, uint32_t *jobs_sorted) try {
    if (n_jobs < 0 || n_keys < 0 || n_ranks < 0) return -1;
    if (n_jobs > 0xFFFFFFFFll) return -1;
    // Stable counting sort by rank, the job list cut into contiguous ranges
    // for T threads: every thread counts the ranks of its range, one serial
    // pass turns the T x n_ranks counts into start positions (rank-major,
    // thread-minor: stable), every thread scatters its range.  The rank of a
    // job is looked up once (pass 1) and kept in 4 bytes per job.
    // (More ranks than fit the caches -- a million class pairs -- still work,
    // only slower: the radix sort of round 3 covered a case no data set of
    // the benchmarks produces and is gone.)
    int T = host_threads(n_jobs, 65536);
    while (T > 1 && (int64_t)T * n_ranks > (int64_t(1) << 24)) --T;
    std::vector<uint32_t> rank((size_t)n_jobs);
    std::vector<std::vector<int64_t>> start((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    parallel_ranges(n_jobs, T, [&](int64_t t0, int64_t t1, int k) {
        std::vector<int64_t> &cnt = start[(size_t)k];
        cnt.assign((size_t)n_ranks, 0);
        for (int64_t t = t0; t < t1; ++t) {
            const int32_t key = pk[t];
            if (key < 0 || key >= n_keys) {
                bad[(size_t)k] = 1;
                return;
            }
            const int32_t r = rank_of_key[key];
            if (r < 0 || r >= n_ranks) {
                bad[(size_t)k] = 1;
                return;
            }
            rank[(size_t)t] = (uint32_t)r;
            ++cnt[(size_t)r];
        }
    });
    for (int k = 0; k < T; ++k)
        if (bad[(size_t)k]) return -1;
    int64_t run = 0;
    for (int64_t r = 0; r < n_ranks; ++r)
        for (int k = 0; k < T; ++k) {
            const int64_t c = start[(size_t)k][(size_t)r];
            start[(size_t)k][(size_t)r] = run;
            run += c;
        }
    const bool with_jobs = jobs && jobs_sorted;
    parallel_ranges(n_jobs, T, [&](int64_t t0, int64_t t1, int k) {
        std::vector<int64_t> &pos = start[(size_t)k];
        for (int64_t t = t0; t < t1; ++t) {
            const int64_t at = pos[rank[(size_t)t]]++;
            order[at] = (uint32_t)t;
            if (with_jobs) {          // the job list in launch order, as uploaded
                jobs_sorted[2 * at] = jobs[2 * t];
                jobs_sorted[2 * at + 1] = jobs[2 * t + 1];
            }
        }
    });
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

int gdh_gather_section(const uint8_t *blob, const int64_t *blob_off, const int64_t *sec_off,
                       int32_t col, const int64_t *count, int64_t G, int32_t itemsize,
                       uint8_t *out, int64_t out_bytes) try {
    if (G < 0 || col < 0 || col >= 6 || itemsize < 0) return -1;
    int64_t at = 0;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t nb = count[g] * (int64_t)itemsize;
        if (nb < 0 || at + nb > out_bytes) return -2;
        std::memcpy(out + at, blob + blob_off[g] + sec_off[g * 6 + col], (size_t)nb);
        at += nb;
    }
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

int gdh_assemble_arena(int64_t G, const uint8_t *blob, const int64_t *blob_off,
                       const int64_t *starts, const int64_t *cbytes, const int64_t *n_node,
                       const int64_t *n_nz, const uint8_t *ncls, const uint8_t *ecls,
                       uint8_t *host, int64_t host_bytes) try {
    if (G < 0) return -1;
    int64_t vn = 0, ve = 0;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t nb = blob_off[g + 1] - blob_off[g];
        if (nb < 0 || starts[g] < cbytes[g] || starts[g] + nb > host_bytes) return -2;
        std::memcpy(host + starts[g], blob + blob_off[g], (size_t)nb);
        if (ncls && ecls && cbytes[g] > 0) {
            uint8_t *c0 = host + starts[g] - cbytes[g];
            const int64_t npad = (n_node[g] + 3) / 4 * 4;
            if (npad + n_nz[g] > cbytes[g]) return -2;
            std::memcpy(c0, ncls + vn, (size_t)n_node[g]);
            std::memcpy(c0 + npad, ecls + ve, (size_t)n_nz[g]);
        }
        vn += n_node[g];
        ve += n_nz[g];
    }
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

int gdh_pairwise_jobs(int64_t nx, int64_t ny, uint32_t *jobs) try {
    if (nx < 0 || ny < -1 || nx + (ny > 0 ? ny : 0) > 0xFFFFFFFFll) return -1;
    int64_t t = 0;
    if (ny < 0) {          // symmetric: the upper triangle with the diagonal
        for (int64_t i = 0; i < nx; ++i)
            for (int64_t j = i; j < nx; ++j, ++t) {
                jobs[2 * t] = (uint32_t)i;
                jobs[2 * t + 1] = (uint32_t)j;
            }
    } else {               // X against Y: the graphs of Y follow those of X
        for (int64_t i = 0; i < nx; ++i)
            for (int64_t j = 0; j < ny; ++j, ++t) {
                jobs[2 * t] = (uint32_t)i;
                jobs[2 * t + 1] = (uint32_t)(nx + j);
            }
    }
    return 0;
} catch (...) { return -3; }   // (bad_alloc, system_error: never across the C ABI)

}  // extern "C"
