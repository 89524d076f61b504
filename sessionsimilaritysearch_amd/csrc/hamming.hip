// Binary-code (Hamming) flat search (gfx950) -- the reference's compressed-index variant:
//   sign bits of the BinarizeHead output (model/model.py:105-138) -> np.packbits((emb + 1) / 2)
//   -> faiss.IndexBinaryFlat(nbits).add / .search(codes, 100)   (fine_tune_ours.py:839-843,871-876)
// SURVEY.md section 8(f) row 3.  Results: D int32 = Hamming distance ascending, I int64; ties are
// ordered by ascending id (faiss' heap order on ties is implementation-defined; the build fixes
// (distance asc, id asc) as it does for the float index).
//
// Byte/integer work, HBM/VALU-bound -- no matrix cores.  k_hamming_scan: one thread per query
// (its code in NW registers), one workgroup per (256 queries, corpus split); 256-row tiles are
// staged through LDS and every thread walks all rows with broadcast ds_reads:
// NW x (v_xor + v_bcnt accumulate) per (query, row).  Tiles are dealt to the splits round-robin,
// so the low ids that win ties are spread over all lists.  Each thread keeps its 16 best
// (distance, id) in registers; k_hamming_select sorts a query's S x 16 candidates and proves
// the result exact (every FULL list's tail is no better than the k-th result) or flags the query
// for the exhaustive path.
#include "sss_common.h"

namespace sss {

constexpr int HK = 16;              // per-thread list length
constexpr int HT_ROWS = 256;        // rows per LDS tile
constexpr int HSEL_THREADS = 256;

template <int N>
__device__ __forceinline__ void hlist_insert(unsigned (&ld)[N], int (&li)[N], unsigned d, int id) {
    // ascending by distance; '<' puts the new row AFTER the entries it ties with (they came earlier = lower ids);
    // from there on every entry moves down one place -- unconditionally: comparing the carried entry again
    // would let it jump over the entries IT ties with and scramble the id order inside a tie
    bool ins = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool c = ins || d < ld[i];
        ins = c;
        const unsigned nd = c ? d : ld[i];
        const int ni = c ? id : li[i];
        d = c ? ld[i] : d;
        id = c ? li[i] : id;
        ld[i] = nd;
        li[i] = ni;
    }
}

template <int NW>       // 32-bit words per code (4, 8, 16 -> 128, 256, 512 bits)
__global__ __launch_bounds__(256) void k_hamming_scan(const unsigned* __restrict__ Q, int nq, const unsigned* __restrict__ C,
                                                      int n, int S, unsigned long long* __restrict__ cand) {
    __shared__ __attribute__((aligned(16))) unsigned tile[2][HT_ROWS * NW];
    const int tid = threadIdx.x;
    const int split = blockIdx.x % S, g = blockIdx.x / S;
    const int q = g * 256 + tid;
    const int q_ld = q < nq ? q : nq - 1;
    unsigned qw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) qw[w] = Q[(size_t)q_ld * NW + w];
    unsigned ld[HK];
    int li[HK];
#pragma unroll
    for (int i = 0; i < HK; ++i) { ld[i] = 0xFFFFFFFFu; li[i] = -1; }
    const int total_tiles = (n + HT_ROWS - 1) / HT_ROWS;
    // register staging of the next tile (thread t loads row t), written to LDS after the barrier
    unsigned stage[NW];
    auto fetch = [&](int t) {
        long row = (long)t * HT_ROWS + tid;
        if (row > n - 1) row = n - 1;
#pragma unroll
        for (int w = 0; w < NW; w += 4)
            *reinterpret_cast<uint4*>(&stage[w]) = *reinterpret_cast<const uint4*>(C + (size_t)row * NW + w);
    };
    int t = split;                                   // tiles split, split + S, ...: round-robin over the splits
    if (t < total_tiles) fetch(t);
    int buf = 0;
    for (; t < total_tiles; t += S) {
#pragma unroll
        for (int w = 0; w < NW; w += 4) *reinterpret_cast<uint4*>(&tile[buf][tid * NW + w]) = *reinterpret_cast<uint4*>(&stage[w]);
        __syncthreads();                             // (double buffered: the previous tile's readers are past their loop)
        if (t + S < total_tiles) fetch(t + S);       // next tile's loads fly under this tile's popcounts
        const int row0 = t * HT_ROWS;
        const int nrows = min(HT_ROWS, n - row0);
        const unsigned* tl = tile[buf];
        auto dist_of = [&](int r) {
            unsigned d = 0;
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                const uint4 c4 = *reinterpret_cast<const uint4*>(tl + r * NW + w);   // same address in every lane: broadcast
                d += __builtin_popcount(c4.x ^ qw[w]) + __builtin_popcount(c4.y ^ qw[w + 1]) +
                     __builtin_popcount(c4.z ^ qw[w + 2]) + __builtin_popcount(c4.w ^ qw[w + 3]);
            }
            return d;
        };
        int r = 0;
        for (; r + 4 <= nrows; r += 4) {             // four rows per trip: the LDS broadcasts overlap the popcounts
            const unsigned d0 = dist_of(r), d1 = dist_of(r + 1), d2 = dist_of(r + 2), d3 = dist_of(r + 3);
            const unsigned dm = min(min(d0, d1), min(d2, d3));
            if (__builtin_amdgcn_ballot_w64(dm < ld[HK - 1]) != 0) {     // rare after the first tiles
                if (d0 < ld[HK - 1]) hlist_insert<HK>(ld, li, d0, row0 + r);
                if (d1 < ld[HK - 1]) hlist_insert<HK>(ld, li, d1, row0 + r + 1);
                if (d2 < ld[HK - 1]) hlist_insert<HK>(ld, li, d2, row0 + r + 2);
                if (d3 < ld[HK - 1]) hlist_insert<HK>(ld, li, d3, row0 + r + 3);
            }
        }
        for (; r < nrows; ++r) {
            const unsigned d = dist_of(r);
            if (d < ld[HK - 1]) hlist_insert<HK>(ld, li, d, row0 + r);
        }
        buf ^= 1;
    }
    if (q < nq) {
        unsigned long long* dst = cand + ((size_t)q * S + split) * HK;
#pragma unroll
        for (int i = 0; i < HK; ++i)
            dst[i] = li[i] >= 0 ? (((unsigned long long)ld[i] << 32) | (unsigned)li[i]) : ~0ull;
    }
}

// One workgroup per query: bitonic sort (ascending) of its S * HK candidate keys, write the first
// k, and prove exactness: a FULL list may have dropped rows, all of them worse than its tail, so
// the result is exact iff no full list's tail beats the k-th result.
__global__ __launch_bounds__(HSEL_THREADS) void k_hamming_select(const unsigned long long* __restrict__ cand, int M, int M2,
                                                                 int k, long id_offset, int* __restrict__ D_out,
                                                                 long* __restrict__ I_out, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long hkeys[];
    __shared__ unsigned long long s_minlast;
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* ck = cand + (size_t)q * M;
    unsigned long long minlast = ~0ull;
    for (int i = tid; i < M2; i += HSEL_THREADS) {
        const unsigned long long v = i < M ? ck[i] : ~0ull;
        hkeys[i] = v;
        if (i < M && (i % HK) == HK - 1 && v != ~0ull && v < minlast) minlast = v;      // tail of a FULL list
    }
    if (tid == 0) s_minlast = ~0ull;
    __syncthreads();
    atomicMin(&s_minlast, minlast);
    for (int kk = 2; kk <= M2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < M2; i += HSEL_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = hkeys[i], b = hkeys[ixj];
                    const bool asc = (i & kk) == 0;
                    if (asc ? a > b : a < b) { hkeys[i] = b; hkeys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += HSEL_THREADS) {
        const unsigned long long v = i < M2 ? hkeys[i] : ~0ull;
        if (v != ~0ull) { D_out[(size_t)q * k + i] = (int)(v >> 32); I_out[(size_t)q * k + i] = (long)(unsigned)v + id_offset; }
        else { D_out[(size_t)q * k + i] = 0x7fffffff; I_out[(size_t)q * k + i] = -1; }      // faiss pads with -1
    }
    if (tid == 0) {
        const unsigned long long kth = k - 1 < M2 ? hkeys[k - 1] : ~0ull;
        status[q] = s_minlast < kth ? 1 : 0;        // (kth == ~0: fewer than k rows exist; then no list can be full and hide more)
        if (kth == ~0ull && s_minlast != ~0ull) status[q] = 1;
    }
}

// ---- exhaustive backstop: distances of selected queries against every row, then k rounds of
// "smallest key above the previous one" (correct for any amount of ties; slow; rare)
template <int NW>
__global__ __launch_bounds__(256) void k_hamming_dists(const unsigned* __restrict__ Q, const int* __restrict__ qsel,
                                                       const unsigned* __restrict__ C, long n, unsigned short* __restrict__ dist) {
    const int f = blockIdx.y;
    unsigned qw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) qw[w] = Q[(size_t)qsel[f] * NW + w];
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long)gridDim.x * 256) {
        unsigned d = 0;
#pragma unroll
        for (int w = 0; w < NW; w += 4) {
            const uint4 c4 = *reinterpret_cast<const uint4*>(C + (size_t)r * NW + w);
            d += __builtin_popcount(c4.x ^ qw[w]) + __builtin_popcount(c4.y ^ qw[w + 1]) +
                 __builtin_popcount(c4.z ^ qw[w + 2]) + __builtin_popcount(c4.w ^ qw[w + 3]);
        }
        dist[(size_t)f * n + r] = (unsigned short)d;
    }
}
__global__ __launch_bounds__(1024) void k_hamming_topk_full(const unsigned short* __restrict__ dist, const int* __restrict__ qsel,
                                                            long n, int k, long id_offset, int* __restrict__ D_out,
                                                            long* __restrict__ I_out) {
    __shared__ unsigned long long wbest[16];
    __shared__ unsigned long long s_prev;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned short* d = dist + (size_t)f * n;
    const size_t out = (size_t)qsel[f] * k;
    unsigned long long prev = 0;
    bool first = true;
    for (int it = 0; it < k; ++it) {
        unsigned long long best = ~0ull;
        for (long i = tid; i < n; i += 1024) {
            const unsigned long long key = ((unsigned long long)d[i] << 32) | (unsigned)i;
            if ((first || key > prev) && key < best) best = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        if (lane == 0) wbest[wv] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long gmin = ~0ull;
            for (int w = 0; w < 16; ++w) gmin = wbest[w] < gmin ? wbest[w] : gmin;
            s_prev = gmin;
            if (gmin == ~0ull) { D_out[out + it] = 0x7fffffff; I_out[out + it] = -1; }
            else { D_out[out + it] = (int)(gmin >> 32); I_out[out + it] = (long)(unsigned)gmin + id_offset; }
        }
        __syncthreads();
        prev = s_prev;
        first = false;
        if (prev == ~0ull) prev = ~0ull - 1;          // exhausted: nothing is > prev any more -> padding
    }
}

// packbits((x + 1) / 2 as int), big-endian bit order, zero padded to whole bytes
// (fine_tune_ours.py:839-840): bit = (int)trunc((x + 1) / 2) != 0.
__global__ __launch_bounds__(256) void k_pack_sign_bits(const float* __restrict__ x, long n, int c, long ldx,
                                                        unsigned char* __restrict__ out, int nbytes) {
    const long total = n * nbytes;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / nbytes;
        const int b = (int)(idx % nbytes);
        unsigned v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = b * 8 + i;
            const int bit = col < c ? ((int)((x[r * ldx + col] + 1.0f) / 2.0f) != 0) : 0;
            v |= (unsigned)bit << (7 - i);
        }
        out[idx] = (unsigned char)v;
    }
}

// ------------------------------------------------------------------------------ host launchers
static int hsplits(long nq, long n) {
    const int G = (int)((nq + 255) / 256);
    int S = 1024 / G;                               // ~4 workgroups per CU
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    const long tiles = (n + HT_ROWS - 1) / HT_ROWS;
    if (S > tiles) S = (int)tiles;
    return S < 1 ? 1 : S;
}
size_t hamming_workspace_bytes(long nq, long n) { return (size_t)nq * hsplits(nq, n) * HK * 8 + 256; }
int hamming_capacity(long nq, long n) { return (nq <= 0 || n <= 0) ? 0 : hsplits(nq, n) * HK; }

template <int NW>
static void launch_hscan(const unsigned* q, int nq, const unsigned* c, int n, int S, unsigned long long* cand, hipStream_t st) {
    const int G = (nq + 255) / 256;
    hipLaunchKernelGGL(k_hamming_scan<NW>, dim3((unsigned)(G * S)), dim3(256), 0, st, q, nq, c, n, S, cand);
}

int hamming_topk(const unsigned char* q, long nq, const unsigned char* codes, long n, int nbytes, int k, long id_offset,
                 int* D_out, long* I_out, int* status, void* ws, size_t ws_bytes, hipStream_t st) {
    if (nq <= 0 || n <= 0 || k <= 0 || (nbytes != 16 && nbytes != 32 && nbytes != 64)) {
        set_error("hamming_topk: need nq, n, k > 0 and 16, 32 or 64 code bytes (got %d)", nbytes);
        return SSS_EINVAL;
    }
    if (n >= (1L << 31) || nq >= (1L << 31)) { set_error("hamming_topk: n and nq must be < 2^31"); return SSS_EINVAL; }
    const int S = hsplits(nq, n);
    const int M = S * HK;
    if (k > M) { set_error("hamming_topk: k %d exceeds the fused capacity %d for this shape (use the exhaustive path)", k, M); return SSS_EINVAL; }
    if (ws_bytes < hamming_workspace_bytes(nq, n)) { set_error("hamming_topk: workspace too small"); return SSS_EWORKSPACE; }
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(ws);
    const unsigned* qu = reinterpret_cast<const unsigned*>(q);
    const unsigned* cu = reinterpret_cast<const unsigned*>(codes);
    if (nbytes == 16) launch_hscan<4>(qu, (int)nq, cu, (int)n, S, cand, st);
    else if (nbytes == 32) launch_hscan<8>(qu, (int)nq, cu, (int)n, S, cand, st);
    else launch_hscan<16>(qu, (int)nq, cu, (int)n, S, cand, st);
    int rc = check_launch("k_hamming_scan");
    if (rc) return rc;
    int M2 = 64;
    while (M2 < M) M2 <<= 1;
    hipLaunchKernelGGL(k_hamming_select, dim3((unsigned)nq), dim3(HSEL_THREADS), (size_t)M2 * 8, st, cand, M, M2, k, id_offset,
                       D_out, I_out, status);
    return check_launch("k_hamming_select");
}

size_t hamming_exhaustive_workspace_bytes(long nsel, long n) { return (size_t)nsel * n * 2 + 256; }

int hamming_topk_exhaustive(const unsigned char* q, const int* qsel, long nsel, const unsigned char* codes, long n, int nbytes,
                            int k, long id_offset, int* D_out, long* I_out, void* ws, size_t ws_bytes, hipStream_t st) {
    if (nsel <= 0 || n <= 0 || k <= 0 || nsel > 65535 || (nbytes != 16 && nbytes != 32 && nbytes != 64)) {
        set_error("hamming_topk_exhaustive: need 0 < nsel <= 65535, n, k > 0 and 16, 32 or 64 code bytes");
        return SSS_EINVAL;
    }
    if (ws_bytes < hamming_exhaustive_workspace_bytes(nsel, n)) { set_error("hamming_topk_exhaustive: workspace too small"); return SSS_EWORKSPACE; }
    unsigned short* dist = reinterpret_cast<unsigned short*>(ws);
    const unsigned* qu = reinterpret_cast<const unsigned*>(q);
    const unsigned* cu = reinterpret_cast<const unsigned*>(codes);
    long gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    const dim3 grid((unsigned)gx, (unsigned)nsel);
    if (nbytes == 16) hipLaunchKernelGGL(k_hamming_dists<4>, grid, dim3(256), 0, st, qu, qsel, cu, n, dist);
    else if (nbytes == 32) hipLaunchKernelGGL(k_hamming_dists<8>, grid, dim3(256), 0, st, qu, qsel, cu, n, dist);
    else hipLaunchKernelGGL(k_hamming_dists<16>, grid, dim3(256), 0, st, qu, qsel, cu, n, dist);
    int rc = check_launch("k_hamming_dists");
    if (rc) return rc;
    hipLaunchKernelGGL(k_hamming_topk_full, dim3((unsigned)nsel), dim3(1024), 0, st, dist, qsel, n, k, id_offset, D_out, I_out);
    return check_launch("k_hamming_topk_full");
}

int pack_sign_bits(const float* x, long n, int c, long ldx, unsigned char* out, int nbytes, hipStream_t st) {
    if (n < 0 || c <= 0 || nbytes * 8 < c || ldx < c) { set_error("pack_sign_bits: need nbytes * 8 >= c and ldx >= c"); return SSS_EINVAL; }
    if (n == 0) return SSS_OK;
    long blocks = (n * nbytes + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pack_sign_bits, dim3((unsigned)blocks), dim3(256), 0, st, x, n, c, ldx, out, nbytes);
    return check_launch("k_pack_sign_bits");
}

}  // namespace sss
