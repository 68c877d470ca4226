// Paced negative-pool selection: exact top-k of Q . Pool^T per query without ever storing the
// [Q, Np] score matrix.
//
// The pool is swept once in row chunks.  Each chunk is scored by the tiled bf16 MFMA kernel of
// gemm.hip (pool rows on M, queries on N) whose epilogue keeps only scores >= the query's current
// threshold (the k-th best score seen so far) and appends them, as order-preserving 64-bit keys
// (score, ~row), to that query's candidate list.  A selection kernel (8-pass byte radix select over
// the keys, then an in-place compaction) trims a list back to its k best whenever it has grown to
// 2k and raises the threshold.  The first chunk is written densely (no atomics) to seed the
// thresholds; chunk sizes double up to a maximum so that the expected number of survivors per chunk
// stays about k.  Candidate capacity 2k + (largest chunk) per query makes overflow impossible.  The largest
// chunk is 131072 rows for many queries and grows to 2^20 rows for few (ch_max_for): with Q below the ridge
// of ~312 queries per pass the sweep is bound by the pool's bytes, a chunk's product takes only ~30 us per
// 131072 rows, and the per-chunk selection launch and the launch gaps would otherwise be half the sweep.  A final
// bitonic sort in LDS orders each query's k keys: (score descending, pool row ascending).
#include "gemm.h"
#include <cstdlib>
#include <mutex>

namespace {

constexpr int CH_FIRST = 4096;
constexpr int CH_MAX = 131072;
constexpr int K_MAX = 4096;

// largest chunk (pool rows per scoring launch) for Q queries: the candidate lists (Q x (2k + chunk) keys of 8 bytes)
// stay within ~0.5 GB
inline int ch_max_for(int Q) { return Q <= 64 ? 1 << 20 : Q <= 128 ? 1 << 19 : Q <= 256 ? 1 << 18 : CH_MAX; }
inline int cap_for(int k, int Q) { return 2 * k + ch_max_for(Q); }

__global__ void topk_init_kernel(unsigned int* cnt, float* thr, int Q, unsigned int cnt0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) { cnt[i] = cnt0; thr[i] = -INFINITY; }
}

// One workgroup per query.  If the list holds >= limit keys, keep exactly its k largest (unordered,
// at the front), set cnt = k and thr = score of the k-th largest.
__global__ void __launch_bounds__(256) topk_select_kernel(unsigned long long* cand_all, unsigned int* cnt_all, float* thr_all,
                                                         int cap, int k, unsigned int limit) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sh_bin, sh_need;
    __shared__ unsigned int wave_off[5];
    const int q = blockIdx.x;
    const unsigned int cnt = min(cnt_all[q], (unsigned int)cap);
    if (cnt < limit || cnt <= (unsigned int)k) return;
    unsigned long long* cand = cand_all + (size_t)q * cap;

    unsigned long long prefix = 0, mask = 0;
    unsigned int need = (unsigned int)k;            // rank (from the top) of the key we are looking for
    for (int pass = 7; pass >= 0; --pass) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (unsigned int i = threadIdx.x; i < cnt; i += 256) {
            const unsigned long long key = cand[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(unsigned int)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // bin of the key of rank `need` from the top: lane l owns bins 4l .. 4l+3; a suffix scan over the lanes
            // finds the one lane whose bins straddle the rank (a serial walk over 256 bins by one thread was most
            // of this kernel's time)
            const int l = threadIdx.x;
            const unsigned int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            const unsigned int own = h0 + h1 + h2 + h3;
            unsigned int suf = own;                                  // sum over lanes >= l
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int up = (unsigned int)__shfl_down((int)suf, o, 64);
                if (l + o < 64) suf += up;
            }
            const unsigned int above = suf - own;                    // keys in higher bins than this lane's
            if (l == 0) { sh_bin = 0; sh_need = need - (suf - h0); }  // rank beyond every bin above 0 (cannot happen: need <= matching keys)
            if (above < need && need <= suf) {
                unsigned int cum = above;
                int b = 4 * l + 3;
                if (cum + h3 < need) { cum += h3; b = 4 * l + 2;
                    if (cum + h2 < need) { cum += h2; b = 4 * l + 1;
                        if (cum + h1 < need) { cum += h1; b = 4 * l; } } }
                sh_bin = (unsigned int)b;
                sh_need = need - cum;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)sh_bin << shift;
        mask |= 0xFFull << shift;
        need = sh_need;
        __syncthreads();
    }
    const unsigned long long kth = prefix;          // keys are unique, so exactly k keys are >= kth

    unsigned int written = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (unsigned int base = 0; base < cnt; base += 256) {
        const unsigned int i = base + threadIdx.x;
        const unsigned long long key = i < cnt ? cand[i] : 0ull;
        const bool keep = i < cnt && key >= kth;
        const unsigned long long bal = __ballot(keep);
        const unsigned int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_off[wv] = __popcll(bal);
        __syncthreads();
        unsigned int off = written;
        for (int w2 = 0; w2 < wv; ++w2) off += wave_off[w2];
        const unsigned int total = wave_off[0] + wave_off[1] + wave_off[2] + wave_off[3];
        if (keep) cand[off + before] = key;          // off + before <= i: never overtakes unread keys
        written += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cnt_all[q] = (unsigned int)k;
        thr_all[q] = topk_key_score(kth);
    }
}

// One workgroup per query: sort the (<= K_MAX) keys at the front of the list descending and emit
// the first k as (score, index).  Unused slots are padded with key 0 (below every real key).
__global__ void __launch_bounds__(256) topk_emit_kernel(const unsigned long long* cand_all, const unsigned int* cnt_all, int cap,
                                                       int k, int npad, long long index_base,
                                                       float* out_scores, long long* out_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    const int q = blockIdx.x;
    const unsigned int cnt = min(cnt_all[q], (unsigned int)npad);
    const unsigned long long* cand = cand_all + (size_t)q * cap;
    for (int i = threadIdx.x; i < npad; i += 256) keys[i] = (unsigned int)i < cnt ? cand[i] : 0ull;
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < npad / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += 256) {
        const unsigned long long key = keys[i];
        const bool real = (unsigned int)i < cnt;
        out_scores[(size_t)q * k + i] = real ? topk_key_score(key) : -INFINITY;
        out_idx[(size_t)q * k + i] = real ? index_base + (long long)topk_key_row(key) : -1;
    }
}

// Merge per-shard results: [Q, total] (score, global index) pairs -> top-k under the same order.
__global__ void __launch_bounds__(256) topk_merge_kernel(const float* scores, const long long* idx, int total, int k, int npad,
                                                        float* out_scores, long long* out_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    const int q = blockIdx.x;
    for (int i = threadIdx.x; i < npad; i += 256) {
        unsigned long long key = 0ull;
        if (i < total) {
            const long long id = idx[(size_t)q * total + i];
            if (id >= 0) key = topk_key(scores[(size_t)q * total + i], (unsigned int)id);
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < npad / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += 256) {
        const unsigned long long key = keys[i];
        out_scores[(size_t)q * k + i] = key ? topk_key_score(key) : -INFINITY;
        out_idx[(size_t)q * k + i] = key ? (long long)topk_key_row(key) : -1;
    }
}

int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

}  // namespace

extern "C" size_t paced_pool_topk_workspace_bytes(int Q, int k) {
    if (Q <= 0 || k <= 0) return 0;
    const size_t head = ((size_t)Q * 8 + 255) / 256 * 256;
    return head + (size_t)Q * cap_for(k, Q) * 8;
}

namespace {

// The sweep itself: candidate lists, counters and thresholds live in memory the caller of this function owns;
// `cap` slots per query, chunks of at most `ch_max` rows (cap >= 2k + ch_max and cap >= CH_FIRST).
int pool_topk_sweep(const void* queries, const void* pool, int Q, long long Np, int D, int k, long long index_base,
                    float* out_scores, long long* out_idx, unsigned int* cnt, float* thr, unsigned long long* cand,
                    int cap, int ch_max, hipStream_t s) {
    GemmParams p = {};
    p.B = (const bf16*)queries; p.N = Q; p.K = D; p.lda = D; p.ldb = D; p.ldc = 0;
    p.k_per_split = D; p.alpha = 1.f; p.mask_scale = 1.f; p.drop = make_drop(0.f, 0); p.drop_narrow = 0;
    p.tk_cand = cand; p.tk_cnt = cnt; p.tk_thr = thr; p.tk_cap = cap;
    p.batch = 1;

    static int stream_env = -1;          // PACED_POOL_STREAM=0: always the tiled kernel (A/B runs)
    if (stream_env < 0) { const char* e = getenv("PACED_POOL_STREAM"); stream_env = e ? atoi(e) : 1; }
    const int stream_maxq = stream_env > 1 ? stream_env : 128;     // PACED_POOL_STREAM=64: the 8-wave form (65..128 queries) off
    const bool stream_ok = stream_env && Q <= stream_maxq && D <= 768 && ((size_t)D * 2 * 128) < 0x7FFFFFFFull;
    // chunk 0: dense write, then select
    long long done = 0;
    int chunk = (int)(Np < CH_FIRST ? Np : CH_FIRST);
    hipLaunchKernelGGL(topk_init_kernel, dim3((Q + 255) / 256), dim3(256), 0, s, cnt, thr, Q, (unsigned int)chunk);
    PACED_CHECK_LAUNCH();
    p.A = (const bf16*)pool; p.M = chunk; p.tk_row_base = 0;
    if (int e = paced_gemm_launch(p, false, false, EPI_TOPK_DENSE, 1, s)) return e;
    hipLaunchKernelGGL(topk_select_kernel, dim3(Q), dim3(256), 0, s, cand, cnt, thr, cap, k, 0u);
    PACED_CHECK_LAUNCH();
    done = chunk;
    while (done < Np) {
        // double until ch_max; with the streaming kernel (a launch costs ~30 us whatever its size, a selection ~30 us
        // more) quadruple: 3k survivors per query and chunk instead of k, half as many launches before the full-size chunks
        long long next = stream_ok ? 3 * done : done;
        if (next > ch_max) next = ch_max;
        if (next > Np - done) next = Np - done;
        p.A = (const bf16*)pool + (size_t)done * D; p.M = (int)next; p.tk_row_base = (unsigned int)done;
        // few queries: the sweep is bound by the pool's bytes and takes the streaming kernel; else the tiled MFMA kernel
        if (!stream_ok || paced_pool_stream_launch(p, s) != PACED_OK)
            if (int e = paced_gemm_launch(p, false, false, EPI_TOPK_FILTER, 1, s)) return e;
        hipLaunchKernelGGL(topk_select_kernel, dim3(Q), dim3(256), 0, s, cand, cnt, thr, cap, k, (unsigned int)(2 * k));
        PACED_CHECK_LAUNCH();
        done += next;
    }
    hipLaunchKernelGGL(topk_select_kernel, dim3(Q), dim3(256), 0, s, cand, cnt, thr, cap, k, 0u);
    PACED_CHECK_LAUNCH();
    const int npad = next_pow2(k);
    hipLaunchKernelGGL(topk_emit_kernel, dim3(Q), dim3(256), (size_t)npad * 8, s, cand, cnt, cap, k, npad, index_base, out_scores, out_idx);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}

}  // namespace

extern "C" int paced_pool_topk(const void* queries, const void* pool, int Q, long long Np, int D, int k,
                               long long index_base, float* out_scores, long long* out_idx,
                               void* workspace, size_t workspace_bytes, void* stream) {
    PACED_CHECK_ARG(queries && pool && out_scores && out_idx && workspace, "paced_pool_topk: null operand");
    PACED_CHECK_ARG(Q > 0 && Np > 0 && D > 0 && D % 64 == 0, "paced_pool_topk: need Q, Np > 0 and D a multiple of 64 (got %d, %lld, %d)", Q, Np, D);
    PACED_CHECK_ARG(k > 0 && k <= K_MAX, "paced_pool_topk: k must be in [1, %d] (got %d)", K_MAX, k);
    PACED_CHECK_ARG(Np < 0xFFFFFFFFll, "paced_pool_topk: at most 2^32-1 pool rows per call (shard larger pools)");
    PACED_CHECK_ARG(workspace_bytes >= paced_pool_topk_workspace_bytes(Q, k), "paced_pool_topk: workspace too small (%zu < %zu)",
                    workspace_bytes, paced_pool_topk_workspace_bytes(Q, k));
    PACED_CHECK_ARG(((uintptr_t)queries & 15) == 0 && ((uintptr_t)pool & 15) == 0 && ((uintptr_t)workspace & 255) == 0, "paced_pool_topk: misaligned buffers");
    unsigned int* cnt = reinterpret_cast<unsigned int*>(workspace);
    float* thr = reinterpret_cast<float*>(cnt + Q);
    const size_t head = ((size_t)Q * 8 + 255) / 256 * 256;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + head);
    return pool_topk_sweep(queries, pool, Q, Np, D, k, index_base, out_scores, out_idx, cnt, thr, cand, cap_for(k, Q), ch_max_for(Q),
                           (hipStream_t)stream);
}

// ---- the same selection for callers that cannot allocate ---------------------------------------------------------
// Candidate lists, counters and thresholds live in one array that is part of the library's device image: every GPU
// the library runs on holds its own copy (STATIC_BYTES, until the process ends) at its own address.  The name
// `scratch` is only a handle in host code; the address the kernels get is looked up once per device with
// hipGetSymbolAddress and kept in g_scratch_ptr.
namespace {

constexpr int STATIC_Q_MAX = 64;            // queries per call
constexpr int STATIC_K_MAX = 1024;          // results per query
constexpr int STATIC_CH_MAX = 32768;        // pool rows per chunk (the sweep above uses CH_MAX = 131072)
constexpr int STATIC_CAP = 2 * STATIC_K_MAX + STATIC_CH_MAX;
constexpr size_t STATIC_HEAD = 1024;        // counters [Q_MAX] u32, thresholds [Q_MAX] f32, padded
constexpr size_t STATIC_BYTES = STATIC_HEAD + (size_t)STATIC_Q_MAX * STATIC_CAP * 8;
constexpr int MAX_DEVICES = 64;

static __device__ __align__(256) unsigned char scratch[STATIC_BYTES];

void* g_scratch_ptr[MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

}  // namespace

extern "C" size_t paced_pool_topk_static_bytes(void) { return STATIC_BYTES; }
extern "C" int paced_pool_topk_static_max_queries(void) { return STATIC_Q_MAX; }
extern "C" int paced_pool_topk_static_max_k(void) { return STATIC_K_MAX; }

extern "C" int paced_pool_topk_static(const void* queries, const void* pool, int Q, long long Np, int D, int k,
                                      long long index_base, float* out_scores, long long* out_idx, void* stream) {
    PACED_CHECK_ARG(queries && pool && out_scores && out_idx, "paced_pool_topk_static: null operand");
    PACED_CHECK_ARG(Q > 0 && Np > 0 && D > 0 && D % 64 == 0, "paced_pool_topk_static: need Q, Np > 0 and D a multiple of 64 (got %d, %lld, %d)", Q, Np, D);
    PACED_CHECK_ARG(Q <= STATIC_Q_MAX, "paced_pool_topk_static: at most %d queries per call (got %d); use paced_pool_topk with a workspace", STATIC_Q_MAX, Q);
    PACED_CHECK_ARG(k > 0 && k <= STATIC_K_MAX, "paced_pool_topk_static: k must be in [1, %d] (got %d); use paced_pool_topk with a workspace", STATIC_K_MAX, k);
    PACED_CHECK_ARG(Np < 0xFFFFFFFFll, "paced_pool_topk_static: at most 2^32-1 pool rows per call (shard larger pools)");
    PACED_CHECK_ARG(((uintptr_t)queries & 15) == 0 && ((uintptr_t)pool & 15) == 0, "paced_pool_topk_static: misaligned buffers");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        paced_set_error("paced_pool_topk_static: no current device (or device index above %d)", MAX_DEVICES - 1);
        return PACED_ERR_HIP;
    }
    void* base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            void* ptr = nullptr;
            const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr) {
                paced_set_error("paced_pool_topk_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
                return PACED_ERR_HIP;
            }
            g_scratch_ptr[dev] = ptr;
        }
        base = g_scratch_ptr[dev];
    }
    // every call starts from counters and thresholds written by topk_init_kernel; stale candidates beyond a
    // query's counter are never read
    unsigned int* cnt = reinterpret_cast<unsigned int*>(base);
    float* thr = reinterpret_cast<float*>(cnt + STATIC_Q_MAX);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(base) + STATIC_HEAD);
    return pool_topk_sweep(queries, pool, Q, Np, D, k, index_base, out_scores, out_idx, cnt, thr, cand, STATIC_CAP, STATIC_CH_MAX,
                           (hipStream_t)stream);
}

extern "C" int paced_topk_merge(const float* scores, const long long* idx, int Q, int total, int k,
                                float* out_scores, long long* out_idx, void* stream) {
    PACED_CHECK_ARG(scores && idx && out_scores && out_idx, "paced_topk_merge: null operand");
    PACED_CHECK_ARG(Q > 0 && total > 0 && k > 0 && k <= total && total <= 8192, "paced_topk_merge: need 0 < k <= total <= 8192 (got k=%d total=%d)", k, total);
    const int npad = next_pow2(total);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(Q), dim3(256), (size_t)npad * 8, (hipStream_t)stream, scores, idx, total, k, npad, out_scores, out_idx);
    PACED_CHECK_LAUNCH();
    return PACED_OK;
}
