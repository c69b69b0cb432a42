// Bag-of-words assignment on the device (C ABI in include/orbv.h): DBoW2 TemplatedVocabulary::transform
// (thirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259) as two kernels.
//
//   k_voc_descend   one thread per descriptor walks the tree.  The tree is re-packed at load time so that the
//                   children of a node are one contiguous run of 32-byte descriptors (a 320-byte streak for k = 10):
//                   per level a thread reads its run with 128-bit loads and keeps the first minimum.
//   k_voc_group     one workgroup per frame turns the per-feature (word, node, weight) triples into the two
//                   std::maps of the reference, flattened in key order: an LDS bitonic sort on (key << 32 | feature)
//                   gives map order and, inside a key, ascending feature index (= push_back order); run heads are
//                   found with a block scan.  BowVector values are summed and normalised in the reference's own
//                   order (sequential doubles), so they are bit-identical.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/orbv.h"
#include "orb_math.h"

int orbx_set_error(int code, const std::string &msg);
hipError_t orbx_lds_opt_in(const void *kernel, size_t bytes); // orbx_api.hip: dynamic LDS above 64 KB, per kernel and per device

#define V_TRY(expr)                                                                                                    \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return orbx_set_error(ORBX_E_NO_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

struct VocDev {
    const int32_t *first;    // [n_nodes + 1] start of node's children in the packed arrays
    const uint32_t *pk_id;   // [n_nodes - 1] node id of each packed child
    const uint4 *pk_desc;    // [n_nodes - 1][2]
    const uint32_t *word_id; // [n_nodes]
    const double *weight;    // [n_nodes]
    int L, n_words, scoring, weighting;
};

struct VocLane {
    hipStream_t stream = nullptr;
    // per-feature scratch of the batch transform
    uint32_t *s_word = nullptr, *s_node = nullptr;
    double *s_w = nullptr;
    size_t s_items = 0;
    // host-pointer staging: ONE device block and its page-locked mirror, laid out for `h_cap` features:
    // [descriptors 32 B][counts n, n_words, n_fv + pad: 16 B][bow ids 4 B][bow values 8 B][fv nodes 4 B][fv offsets 4 B (+1)][fv indices 4 B]
    uint8_t *d_blk = nullptr, *h_blk = nullptr;
    size_t h_cap = 0;
    static size_t al(size_t x) { return (x + 15) & ~(size_t)15; }
    size_t o_cnt() const { return al(h_cap * 32); }
    size_t o_ids() const { return o_cnt() + 16; }
    size_t o_vals() const { return al(o_ids() + h_cap * 4); }
    size_t o_nodes() const { return o_vals() + h_cap * 8; }
    size_t o_off() const { return al(o_nodes() + h_cap * 4); }
    size_t o_idx() const { return al(o_off() + (h_cap + 1) * 4); }
    size_t bytes() const { return al(o_idx() + h_cap * 4); }
    void release()
    {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); stream = nullptr; }
        for (void *p : {(void *)s_word, (void *)s_node, (void *)s_w, (void *)d_blk})
            if (p) (void)hipFree(p);
        if (h_blk) (void)hipHostFree(h_blk);
        s_word = s_node = nullptr; s_w = nullptr; s_items = 0;
        d_blk = h_blk = nullptr; h_cap = 0;
    }
};

struct orbv_ctx {
    int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0;
    std::vector<int32_t> parent;
    std::vector<uint8_t> is_leaf, desc;
    std::vector<double> weight;
    VocDev dev{};
    void *d_first = nullptr, *d_pk_id = nullptr, *d_pk_desc = nullptr, *d_word_id = nullptr, *d_weight = nullptr;
    bool null_pending = false; // a device call was enqueued on stream 0 (NULL): destroy waits for it too
    // The reference's vocabulary is ONE object that Tracking (Frame::computeBow, Frame.cpp:168-178) and LocalMapping
    // (KeyFrame::computeBow, LocalMapping.cpp:90) call at the same time: the tree above is read-only, everything a call writes
    // lives in a lane.  The device entry points use `dev_lane` (scratch only; one call in flight per handle, as orbv.h says);
    // every host-pointer call leases a lane of its own -- non-blocking stream, scratch, staging -- so orbv_transform is re-entrant.
    VocLane dev_lane;
    std::mutex mu;
    std::vector<VocLane *> idle;
};

__device__ __forceinline__ int ham256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// transform(feature, word, weight, &nid, levelsup), TemplatedVocabulary.h:1218-1259.  parent[i] < i is checked at load
// time, so every descent reaches a childless node and the loop ends.
__global__ __launch_bounds__(256) void k_voc_descend(VocDev v, const uint8_t *__restrict__ desc,
                                                     const int32_t *__restrict__ n_per_frame, int n_single, int cap,
                                                     int levelsup, uint32_t *__restrict__ out_word,
                                                     uint32_t *__restrict__ out_node, double *__restrict__ out_w)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_per_frame ? min(n_per_frame[f], cap) : n_single;
    if (i >= n) return;
    const size_t slot = (size_t)f * cap + i;
    const uint4 *dp = reinterpret_cast<const uint4 *>(desc + slot * 32);
    const uint4 a0 = dp[0], a1 = dp[1];
    const int nid_level = v.L - levelsup;
    uint32_t nid = nid_level <= 0 ? 0u : ORBV_NO_NODE; // :1228
    int node = 0, level = 0;
    int b = v.first[0], e = v.first[1];
    do {
        ++level;
        int best = 0x7fffffff, arg = b;
        // children in groups of five whose ten 128-bit loads are requested together (index clamped, no branch around the
        // loads: a loop with a dynamic trip count would otherwise cost one memory latency per child)
        for (int c0 = b; c0 < e; c0 += 5) {
            uint4 lo[5], hi[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const size_t c = (size_t)min(c0 + u, e - 1);
                lo[u] = v.pk_desc[2 * c];
                hi[u] = v.pk_desc[2 * c + 1];
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int d = ham256(a0, a1, lo[u], hi[u]);
                if (c0 + u < e && d < best) { // strict: the first child keeps ties (:1241)
                    best = d;
                    arg = c0 + u;
                }
            }
        }
        node = (int)v.pk_id[arg];
        if (level == nid_level) nid = (uint32_t)node;
        b = v.first[node];
        e = v.first[node + 1];
    } while (b != e); // !isLeaf()
    out_word[slot] = v.word_id[node];
    out_node[slot] = nid;
    out_w[slot] = v.weight[node];
}

// LDS bitonic sort of P (power of two) 64-bit keys with 256 threads
__device__ void bitonic_sort(unsigned long long *keys, int P)
{
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += 256) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long x = keys[lo], y = keys[hi];
                if ((x > y) == up) {
                    keys[lo] = y;
                    keys[hi] = x;
                }
            }
            __syncthreads();
        }
}

// exclusive block scan of one int per thread (256 threads); returns the exclusive prefix, *total = sum
__device__ int block_scan_256(int v, int *wave_sum, int *total)
{
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((threadIdx.x & 63) >= o) incl += t;
    }
    __syncthreads(); // wave_sum may still be read from a previous call
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_sum[w];
    *total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    return base + incl - v;
}

// Sort (key32 << 32 | feature) for the valid features of a frame and find the runs of equal key32.
// On return keys[0..m) are sorted, and for every run r: run_key/run_start are delivered through `emit`.
template <typename Emit>
__device__ int sort_and_runs(unsigned long long *keys, int n, int P, const uint32_t *key32, const double *w, int *wave_sum,
                             int *m_out, Emit emit)
{
    for (int t = threadIdx.x; t < P; t += 256)
        keys[t] = (t < n && w[t] > 0) ? ((unsigned long long)key32[t] << 32) | (unsigned)t : ~0ull; // w > 0: not stopped
    __syncthreads();
    bitonic_sort(keys, P);
    // valid entries first (a real key never equals ~0: feature index < 2^32 - 1)
    const int per = (P + 255) / 256, t0 = min((int)threadIdx.x * per, P), t1 = min(t0 + per, P);
    int heads = 0, valid = 0;
    for (int t = t0; t < t1; ++t) {
        const unsigned long long kx = keys[t];
        if (kx == ~0ull) break;
        ++valid;
        heads += t == 0 || (uint32_t)(keys[t - 1] >> 32) != (uint32_t)(kx >> 32);
    }
    int m, R;
    (void)block_scan_256(valid, wave_sum, &m);
    int run = block_scan_256(heads, wave_sum, &R);
    for (int t = t0; t < t1; ++t) {
        const unsigned long long kx = keys[t];
        if (kx == ~0ull) break;
        if (t == 0 || (uint32_t)(keys[t - 1] >> 32) != (uint32_t)(kx >> 32)) emit(run++, (uint32_t)(kx >> 32), t, (uint32_t)kx);
    }
    *m_out = m;
    return R;
}

__global__ __launch_bounds__(256) void k_voc_group(VocDev v, const int32_t *__restrict__ n_per_frame, int cap, int p_max,
                                                   const uint32_t *__restrict__ word, const uint32_t *__restrict__ node,
                                                   const double *__restrict__ w, uint32_t *__restrict__ bow_ids,
                                                   double *__restrict__ bow_vals, int32_t *__restrict__ n_words,
                                                   uint32_t *__restrict__ fv_nodes, int32_t *__restrict__ fv_off,
                                                   uint32_t *__restrict__ fv_idx, int32_t *__restrict__ n_fv)
{
    extern __shared__ unsigned long long lds64[]; // keys[p_max], vals[p_max]
    __shared__ int wave_sum[4];
    __shared__ double norm_sh;
    unsigned long long *keys = lds64;
    double *vals = reinterpret_cast<double *>(lds64 + p_max);
    int2 *heads = reinterpret_cast<int2 *>(vals); // (run start, feature of the run's head) until the values replace them
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(min(n_per_frame[f], cap), ORBV_MAX_FEATURES);
    word += (size_t)f * cap; node += (size_t)f * cap; w += (size_t)f * cap;
    bow_ids += (size_t)f * cap; bow_vals += (size_t)f * cap;
    fv_nodes += (size_t)f * cap; fv_off += (size_t)f * (cap + 1); fv_idx += (size_t)f * cap;
    if (v.n_words == 0 || n == 0) { // empty() (:1133) / no features: both maps empty
        if (tid == 0) { n_words[f] = 0; n_fv[f] = 0; fv_off[0] = 0; }
        return;
    }
    int P = 1;
    while (P < n) P <<= 1;

    // ---- FeatureVector: map<NodeId, vector<feature>> (FeatureVector.cpp:31-45)
    int m;
    const int n_runs = sort_and_runs(keys, n, P, node, w, wave_sum, &m,
                                     [&](int run, uint32_t key, int start, uint32_t) { fv_nodes[run] = key; fv_off[run] = start; });
    for (int t = tid; t < m; t += 256) fv_idx[t] = (uint32_t)keys[t];
    if (tid == 0) { fv_off[n_runs] = m; n_fv[f] = n_runs; }
    __syncthreads();

    // ---- BowVector: map<WordId, double> (BowVector.cpp:32-45), then the normalisation of :1164-1201
    const int R = sort_and_runs(keys, n, P, word, w, wave_sum, &m,
                                [&](int run, uint32_t key, int start, uint32_t feat) { bow_ids[run] = key; heads[run] = make_int2(start, (int)feat); });
    __syncthreads();
    const bool accumulate = v.weighting == ORBV_TF_IDF || v.weighting == ORBV_TF;
    for (int r0 = 0; r0 < R; r0 += 256) { // heads[] and vals[] share storage: read a batch, barrier, write it
        const int r = r0 + tid;
        double val = 0;
        if (r < R) {
            const int2 h = heads[r];
            const int cnt = (r + 1 < R ? heads[r + 1].x : m) - h.x;
            const double wt = w[h.y];
            val = wt;
            if (accumulate)
                for (int j = 1; j < cnt; ++j) val = ORB_DADD(val, wt); // addWeight, one feature after the other
        }
        __syncthreads();
        if (r < R) vals[r] = val;
    }
    __syncthreads();
    const bool must = v.scoring != ORBV_DOT_PRODUCT, l2 = v.scoring == ORBV_L2_NORM;
    if (tid == 0) { // sequential, in map order, as BowVector::normalize sums (BowVector.cpp:67-78)
        double norm = 0.0;
        if (!must) norm = accumulate ? (double)R : 1.0; // :1164-1170 divides by v.size()
        else if (!l2) for (int r = 0; r < R; ++r) norm = ORB_DADD(norm, fabs(vals[r]));
        else {
            for (int r = 0; r < R; ++r) norm = ORB_DADD(norm, ORB_DMUL(vals[r], vals[r]));
            norm = __dsqrt_rn(norm);
        }
        norm_sh = norm;
        n_words[f] = R;
    }
    __syncthreads();
    const double norm = norm_sh;
    const bool divide = must ? norm > 0.0 : accumulate;
    for (int r = tid; r < R; r += 256) bow_vals[r] = divide ? ORB_DDIV(vals[r], norm) : vals[r];
}

// ---------------------------------------------------------------------------------------------------------------------
static int upload(orbv_ctx *c)
{
    const int n = c->n_nodes;
    std::vector<int32_t> first((size_t)n + 1, 0);
    for (int i = 1; i < n; ++i) first[c->parent[i] + 1]++;
    for (int i = 0; i < n; ++i) first[i + 1] += first[i];
    std::vector<int32_t> fill(first.begin(), first.end() - 1);
    std::vector<uint32_t> pk_id((size_t)std::max(n - 1, 1));
    std::vector<uint8_t> pk_desc((size_t)std::max(n - 1, 1) * 32);
    std::vector<uint32_t> word_id((size_t)n, 0); // Node(): word_id(0)
    int n_words = 0;
    for (int i = 1; i < n; ++i) { // ascending id = push_back order (:1393)
        const int slot = fill[c->parent[i]]++;
        pk_id[slot] = (uint32_t)i;
        memcpy(&pk_desc[(size_t)slot * 32], &c->desc[(size_t)i * 32], 32);
        if (c->is_leaf[i]) word_id[i] = (uint32_t)n_words++; // :1408-1414
    }
    c->n_words = n_words;
    V_TRY(hipMalloc(&c->d_first, first.size() * 4));
    V_TRY(hipMalloc(&c->d_pk_id, pk_id.size() * 4));
    V_TRY(hipMalloc(&c->d_pk_desc, pk_desc.size()));
    V_TRY(hipMalloc(&c->d_word_id, word_id.size() * 4));
    V_TRY(hipMalloc(&c->d_weight, (size_t)n * 8));
    V_TRY(hipMemcpy(c->d_first, first.data(), first.size() * 4, hipMemcpyHostToDevice));
    V_TRY(hipMemcpy(c->d_pk_id, pk_id.data(), pk_id.size() * 4, hipMemcpyHostToDevice));
    V_TRY(hipMemcpy(c->d_pk_desc, pk_desc.data(), pk_desc.size(), hipMemcpyHostToDevice));
    V_TRY(hipMemcpy(c->d_word_id, word_id.data(), word_id.size() * 4, hipMemcpyHostToDevice));
    V_TRY(hipMemcpy(c->d_weight, c->weight.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    c->dev = VocDev{(const int32_t *)c->d_first, (const uint32_t *)c->d_pk_id, (const uint4 *)c->d_pk_desc,
                    (const uint32_t *)c->d_word_id, (const double *)c->d_weight, c->L, c->n_words, c->scoring, c->weighting};
    return ORBX_OK;
}

extern "C" int orbv_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                           const uint8_t *is_leaf, const uint8_t *desc, const double *weight, int device, orbv_t **out)
{
    if (!out || n_nodes < 1 || (n_nodes > 1 && (!parent || !is_leaf || !desc || !weight)))
        return orbx_set_error(ORBX_E_ARG, "null argument or empty node list");
    // the ranges loadFromTextFile accepts (:1360)
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
        return orbx_set_error(ORBX_E_ARG, "vocabulary header out of range (k 0..20, L 1..10, scoring 0..5, weighting 0..3)");
    for (int i = 1; i < n_nodes; ++i)
        if (parent[i] < 0 || parent[i] >= i)
            return orbx_set_error(ORBX_E_ARG, "node " + std::to_string(i) + ": parent must be an earlier node");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return orbx_set_error(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return orbx_set_error(ORBX_E_ARG, "device index out of range");
    V_TRY(hipSetDevice(device));
    orbv_ctx *c = new orbv_ctx();
    c->device = device; c->k = k; c->L = L; c->scoring = scoring; c->weighting = weighting; c->n_nodes = n_nodes;
    c->parent.assign((size_t)n_nodes, 0);
    c->is_leaf.assign((size_t)n_nodes, 0);
    c->desc.assign((size_t)n_nodes * 32, 0);
    c->weight.assign((size_t)n_nodes, 0.0); // Node(): weight(0)
    if (n_nodes > 1) {
        memcpy(c->parent.data() + 1, parent + 1, (size_t)(n_nodes - 1) * 4);
        memcpy(c->is_leaf.data() + 1, is_leaf + 1, (size_t)(n_nodes - 1));
        memcpy(c->desc.data() + 32, desc + 32, (size_t)(n_nodes - 1) * 32);
        memcpy(c->weight.data() + 1, weight + 1, (size_t)(n_nodes - 1) * 8);
    }
    int rc = upload(c);
    if (rc) {
        orbv_destroy(c);
        return rc;
    }
    *out = c;
    return ORBX_OK;
}

extern "C" int orbv_load_text(const char *path, int device, orbv_t **out)
{
    if (!path || !out) return orbx_set_error(ORBX_E_ARG, "null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return orbx_set_error(ORBX_E_ARG, std::string("cannot open vocabulary ") + path + ": " + strerror(errno));
    std::string txt;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) txt.append(buf, got);
    fclose(fp);
    const char *p = txt.c_str(), *end = p + txt.size();
    auto line_end = [&](const char *q) { while (q < end && *q != '\n') ++q; return q; };
    // header "k L scoring weighting" (:1352-1366)
    const char *le = line_end(p);
    long hdr[4];
    {
        std::string h(p, le);
        char *q = &h[0];
        for (int i = 0; i < 4; ++i) {
            char *nx;
            hdr[i] = strtol(q, &nx, 10);
            if (nx == q) return orbx_set_error(ORBX_E_ARG, "vocabulary header: expected 'k L scoring weighting'");
            q = nx;
        }
    }
    if (hdr[0] < 0 || hdr[0] > 20 || hdr[1] < 1 || hdr[1] > 10 || hdr[2] < 0 || hdr[2] > 5 || hdr[3] < 0 || hdr[3] > 3)
        return orbx_set_error(ORBX_E_ARG, "Vocabulary loading failure: This is not a correct text file!"); // :1362
    std::vector<int32_t> parent(1, 0);
    std::vector<uint8_t> is_leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    p = le < end ? le + 1 : end;
    long line_no = 1;
    while (p < end) {
        le = line_end(p);
        ++line_no;
        const char *q = p;
        while (q < le && (*q == ' ' || *q == '\r' || *q == '\t')) ++q;
        if (q == le) { // blank line: the reference would append a phantom node here (see orbv.h); we do not
            p = le < end ? le + 1 : end;
            continue;
        }
        std::string ln(p, le);
        char *c = &ln[0], *nx;
        const long pid = strtol(c, &nx, 10);
        bool ok = nx != c;
        c = nx;
        const long leaf = strtol(c, &nx, 10);
        ok = ok && nx != c;
        c = nx;
        uint8_t d[32];
        for (int i = 0; i < 32 && ok; ++i) {
            const long b = strtol(c, &nx, 10);
            ok = nx != c;
            c = nx;
            d[i] = (uint8_t)b; // (unsigned char)n, FORB.cpp:133
        }
        const double wgt = ok ? strtod(c, &nx) : 0.0;
        ok = ok && nx != c;
        const long nid = (long)parent.size();
        if (!ok || pid < 0 || pid >= nid)
            return orbx_set_error(ORBX_E_ARG, "vocabulary line " + std::to_string(line_no) + ": malformed node record");
        parent.push_back((int32_t)pid);
        is_leaf.push_back(leaf > 0); // :1408
        desc.insert(desc.end(), d, d + 32);
        weight.push_back(wgt);
        p = le < end ? le + 1 : end;
    }
    return orbv_create((int)hdr[0], (int)hdr[1], (int)hdr[2], (int)hdr[3], (int)parent.size(), parent.data(), is_leaf.data(),
                       desc.data(), weight.data(), device, out);
}

extern "C" void orbv_destroy(orbv_t *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->null_pending) (void)hipStreamSynchronize((hipStream_t)0);
    for (VocLane *ln : c->idle) { ln->release(); delete ln; }
    c->dev_lane.release();
    for (void *p : {c->d_first, c->d_pk_id, c->d_pk_desc, c->d_word_id, c->d_weight})
        if (p) (void)hipFree(p);
    delete c;
}

extern "C" int orbv_info(const orbv_t *c, int *k, int *L, int *scoring, int *weighting, int *n_nodes, int *n_words)
{
    if (!c) return orbx_set_error(ORBX_E_ARG, "null handle");
    if (k) *k = c->k;
    if (L) *L = c->L;
    if (scoring) *scoring = c->scoring;
    if (weighting) *weighting = c->weighting;
    if (n_nodes) *n_nodes = c->n_nodes;
    if (n_words) *n_words = c->n_words;
    return ORBX_OK;
}

extern "C" int orbv_nodes(const orbv_t *c, int32_t *parent, uint8_t *is_leaf, uint8_t *desc, double *weight)
{
    if (!c) return orbx_set_error(ORBX_E_ARG, "null handle");
    if (parent) memcpy(parent, c->parent.data(), c->parent.size() * 4);
    if (is_leaf) memcpy(is_leaf, c->is_leaf.data(), c->is_leaf.size());
    if (desc) memcpy(desc, c->desc.data(), c->desc.size());
    if (weight) memcpy(weight, c->weight.data(), c->weight.size() * 8);
    return ORBX_OK;
}

static int check_levelsup(int levelsup)
{
    if (levelsup < 0) return orbx_set_error(ORBX_E_ARG, "levelsup must be >= 0");
    return ORBX_OK;
}

extern "C" int orbv_transform_features_device(orbv_t *c, const uint8_t *d_desc, int n, int levelsup, uint32_t *d_word,
                                              uint32_t *d_node, double *d_weight, void *stream)
{
    if (!c || !d_desc || !d_word || !d_node || !d_weight) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n < 0) return orbx_set_error(ORBX_E_ARG, "negative feature count");
    if (check_levelsup(levelsup)) return ORBX_E_ARG;
    if (c->n_words == 0) return orbx_set_error(ORBX_E_ARG, "empty vocabulary");
    if (n == 0) return ORBX_OK;
    V_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    hipLaunchKernelGGL(k_voc_descend, dim3((n + 255) / 256, 1), dim3(256), 0, s, c->dev, d_desc, (const int32_t *)nullptr, n,
                       n, levelsup, d_word, d_node, d_weight);
    V_TRY(hipGetLastError());
    return ORBX_OK;
}

// k_voc_descend + k_voc_group of a batch on stream `s`, with lane `ln`'s per-feature scratch
static int transform_enqueue(orbv_ctx *c, VocLane &ln, int n_frames, const uint8_t *d_desc, const int32_t *d_n, int cap, int levelsup,
                             uint32_t *d_bow_ids, double *d_bow_vals, int32_t *d_n_words, uint32_t *d_fv_nodes, int32_t *d_fv_off,
                             uint32_t *d_fv_idx, int32_t *d_n_fv, hipStream_t s)
{
    const size_t need = (size_t)n_frames * cap;
    if (need > ln.s_items) {
        V_TRY(hipDeviceSynchronize());
        for (void **p : {(void **)&ln.s_word, (void **)&ln.s_node, (void **)&ln.s_w})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        ln.s_items = 0;
        const size_t grow = need + need / 2;
        V_TRY(hipMalloc(&ln.s_word, grow * 4));
        V_TRY(hipMalloc(&ln.s_node, grow * 4));
        V_TRY(hipMalloc(&ln.s_w, grow * 8));
        ln.s_items = grow;
    }
    if (c->n_words > 0) { // empty vocabulary: transform() returns empty maps (:1133), k_voc_group handles it
        hipLaunchKernelGGL(k_voc_descend, dim3((std::min(cap, ORBV_MAX_FEATURES) + 255) / 256, n_frames), dim3(256), 0, s,
                           c->dev, d_desc, d_n, 0, cap, levelsup, ln.s_word, ln.s_node, ln.s_w);
        V_TRY(hipGetLastError());
    }
    int p_max = 256;
    while (p_max < std::min(cap, ORBV_MAX_FEATURES)) p_max <<= 1;
    const size_t lds = (size_t)p_max * 16;
    // more than 64 KB of dynamic LDS has to be requested once per device
    V_TRY(orbx_lds_opt_in(reinterpret_cast<const void *>(k_voc_group), (size_t)ORBV_MAX_FEATURES * 16));
    hipLaunchKernelGGL(k_voc_group, dim3(n_frames), dim3(256), lds, s, c->dev, d_n, cap, p_max, ln.s_word, ln.s_node, ln.s_w,
                       d_bow_ids, d_bow_vals, d_n_words, d_fv_nodes, d_fv_off, d_fv_idx, d_n_fv);
    V_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbv_transform_device(orbv_t *c, int n_frames, const uint8_t *d_desc, const int32_t *d_n, int cap,
                                     int levelsup, uint32_t *d_bow_ids, double *d_bow_vals, int32_t *d_n_words,
                                     uint32_t *d_fv_nodes, int32_t *d_fv_off, uint32_t *d_fv_idx, int32_t *d_n_fv,
                                     void *stream)
{
    if (!c || !d_desc || !d_n || !d_bow_ids || !d_bow_vals || !d_n_words || !d_fv_nodes || !d_fv_off || !d_fv_idx || !d_n_fv)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n_frames < 0 || cap <= 0) return orbx_set_error(ORBX_E_ARG, "n_frames must be >= 0 and cap positive");
    // the counts live on the device, so a frame with more features than the grouping kernel can sort cannot be
    // reported from here: refuse a capacity that would allow one, like the host path refuses the count itself
    if (cap > ORBV_MAX_FEATURES)
        return orbx_set_error(ORBX_E_UNSUPPORTED, "cap exceeds ORBV_MAX_FEATURES features per frame");
    if (check_levelsup(levelsup)) return ORBX_E_ARG;
    if (n_frames == 0) return ORBX_OK;
    V_TRY(hipSetDevice(c->device));
    if (!stream) c->null_pending = true; // NULL is stream 0 itself (include/orbx.h, "Streams")
    return transform_enqueue(c, c->dev_lane, n_frames, d_desc, d_n, cap, levelsup, d_bow_ids, d_bow_vals, d_n_words, d_fv_nodes, d_fv_off,
                             d_fv_idx, d_n_fv, (hipStream_t)stream);
}

namespace {
// a host-pointer call's lane: taken from the handle's idle list (or made), given back when the call returns
struct LaneLease {
    orbv_ctx *c;
    VocLane *ln = nullptr;
    explicit LaneLease(orbv_ctx *ctx) : c(ctx) {}
    ~LaneLease()
    {
        if (!ln) return;
        (void)hipStreamSynchronize(ln->stream); // an error return may leave work in flight
        std::lock_guard<std::mutex> lock(c->mu);
        c->idle.push_back(ln);
    }
    hipError_t acquire()
    {
        {
            std::lock_guard<std::mutex> lock(c->mu);
            if (!c->idle.empty()) { ln = c->idle.back(); c->idle.pop_back(); return hipSuccess; }
        }
        VocLane *n = new VocLane();
        hipError_t e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete n; return e; }
        ln = n;
        return hipSuccess;
    }
};
} // namespace

extern "C" int orbv_transform(orbv_t *c, const uint8_t *desc, int n, int levelsup, uint32_t *bow_ids, double *bow_vals,
                              int32_t *n_words, uint32_t *fv_nodes, int32_t *fv_off, uint32_t *fv_idx, int32_t *n_fv)
{
    if (!c || !bow_ids || !bow_vals || !n_words || !fv_nodes || !fv_off || !fv_idx || !n_fv || (n > 0 && !desc))
        return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n < 0) return orbx_set_error(ORBX_E_ARG, "negative feature count");
    if (n > ORBV_MAX_FEATURES) return orbx_set_error(ORBX_E_UNSUPPORTED, "more than ORBV_MAX_FEATURES features in one frame");
    *n_words = 0;
    *n_fv = 0;
    fv_off[0] = 0;
    if (n == 0) return ORBX_OK;
    if (check_levelsup(levelsup)) return ORBX_E_ARG;
    V_TRY(hipSetDevice(c->device));
    LaneLease lease(c);
    V_TRY(lease.acquire());
    VocLane &ln = *lease.ln;
    hipStream_t s = ln.stream;
    if ((size_t)n > ln.h_cap) {
        V_TRY(hipStreamSynchronize(s));
        if (ln.d_blk) { (void)hipFree(ln.d_blk); ln.d_blk = nullptr; }
        if (ln.h_blk) { (void)hipHostFree(ln.h_blk); ln.h_blk = nullptr; }
        ln.h_cap = (size_t)n + n / 2 + 64;
        const size_t nb = ln.bytes();
        hipError_t e = hipMalloc((void **)&ln.d_blk, nb);
        if (e == hipSuccess) e = hipHostMalloc((void **)&ln.h_blk, nb, hipHostMallocDefault);
        if (e != hipSuccess) { ln.h_cap = 0; V_TRY(e); }
    }
    // one copy up (descriptors + the count), the two kernels, one copy down (counts and the five arrays at their capacity), ONE wait:
    // the arrays then go to the caller from page-locked memory (was two waits and eight copies, five of them to pageable memory)
    uint8_t *d = ln.d_blk, *h = ln.h_blk;
    memcpy(h, desc, (size_t)n * 32);
    int32_t *hc = reinterpret_cast<int32_t *>(h + ln.o_cnt());
    hc[0] = n; hc[1] = 0; hc[2] = 0; hc[3] = 0;
    V_TRY(hipMemcpyAsync(d, h, ln.o_ids(), hipMemcpyHostToDevice, s));
    int32_t *cnt = reinterpret_cast<int32_t *>(d + ln.o_cnt()); // [0] n, [1] n_words, [2] n_fv
    int rc = transform_enqueue(c, ln, 1, d, cnt, n, levelsup, reinterpret_cast<uint32_t *>(d + ln.o_ids()), reinterpret_cast<double *>(d + ln.o_vals()),
                               cnt + 1, reinterpret_cast<uint32_t *>(d + ln.o_nodes()), reinterpret_cast<int32_t *>(d + ln.o_off()),
                               reinterpret_cast<uint32_t *>(d + ln.o_idx()), cnt + 2, s);
    if (rc) return rc;
    // (the arrays are laid out for h_cap features; only the first n -- n + 1 offsets -- of each can hold anything)
    const size_t down_end = ln.o_idx() + (size_t)n * 4;
    V_TRY(hipMemcpyAsync(h + ln.o_cnt(), d + ln.o_cnt(), down_end - ln.o_cnt(), hipMemcpyDeviceToHost, s));
    V_TRY(hipStreamSynchronize(s));
    const int nw = hc[1], nf = hc[2];
    if (nw < 0 || nw > n || nf < 0 || nf > n) return orbx_set_error(ORBX_E_NO_DEVICE, "orbv_transform: counts out of range");
    *n_words = nw;
    *n_fv = nf;
    memcpy(bow_ids, h + ln.o_ids(), (size_t)nw * 4);
    memcpy(bow_vals, h + ln.o_vals(), (size_t)nw * 8);
    memcpy(fv_nodes, h + ln.o_nodes(), (size_t)nf * 4);
    memcpy(fv_off, h + ln.o_off(), (size_t)(nf + 1) * 4);
    memcpy(fv_idx, h + ln.o_idx(), (size_t)n * 4);
    return ORBX_OK;
}
