// Query x corpus inner-product CANDIDATE scan with fused running top-k (gfx950 / CDNA4): f32, bf16,
// split-bf16 (three passes over an f32 corpus) and scaled f16 (one pass over an f32 corpus).
//
// Replaces what the reference asks of faiss at test_amazon_filterd.py:578
// (`D, I = index.search(normalize(emb), K)`, faiss.IndexFlatIP).
// The scan's scores never reach the output: select.hip re-scores the candidates in float64 from the
// stored rows and proves, per query, that nothing outside them could matter (error bound per scan
// type: select.hip err_bound).
//
// k_scan<RB, TR, DT, NW, THR, AP>  -- the dominant kernel (DESIGN.md "scoring kernel"); RB = bytes per scanned
// corpus row (256 / 512 / 1024), TR = rows per LDS tile (64 / 128 / 256), DT = element type (scan.h),
// NW = waves per workgroup:
//   * one workgroup = 8 waves (2 per SIMD) = 256 queries x one contiguous corpus split (1024-byte
//     rows: 4 waves, one per SIMD, 128 queries);
//   * each wave keeps its 32 queries resident in RB/8 VGPRs as the B operand of
//     v_mfma_f32_32x32x2_f32 (DT_F32: an exact k-ordered f32 fma chain), v_mfma_f32_32x32x16_bf16
//     (DT_BF16; DT_SPLIT: hi*hi + hi*lo + lo*hi of rows stored [hi | lo]) or v_mfma_f32_32x32x16_f16
//     (DT_F16: f32 queries scaled by their own power of two and rounded in the prologue), so the
//     query tile is read from HBM once;
//   * corpus rows stream HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging), double
//     buffered, one burst per tile, 16-byte chunks XOR-swizzled on the SOURCE address so the
//     ds_read_b128 fragment reads are bank-conflict free.  Rows of equal BYTES stage and read
//     identically whatever they hold: lane half h reads chunk 2u+h, which is the k-permuted A operand
//     of four f32 MFMAs or the natural A operand of one 16-bit MFMA;
//   * the score matrix is never written: each lane owns one query column of the 32x32
//     accumulators and keeps a sorted top-KP list (scores + row ids) in registers.  The scan runs
//     in 64-row steps: a HOT loop (MFMAs, one v_max3 tree, one compare) that never writes the list
//     state, left for a RARE insert path only when some lane's step maximum beats its threshold;
//   * ADMISSION THRESHOLD shared by all workgroups of a query: every lane list belongs to one of J
//     classes (J >= K2); slot[q][class] holds, by atomic max, the best score any list of that class
//     has seen (cert == 1) or the largest cert-th best of such a list (cert > 1).  Classes
//     partition the corpus rows, so tau = min over slots is a score that at least J * cert >= K2
//     distinct rows reach, and a row scoring below tau can never be among the best K2.  With
//     cert == 1 (K2 <= 16) all 16 slots are distinct classes and tau is RANK-SELECTED: the K2-th largest
//     class maximum (scan_dev.h: tau_select16) -- K2 distinct rows reach it as well, and it sits near the
//     ~21st best row seen instead of the ~37th of "min over K2 classes".  The first tile of every split is
//     scanned twice: once max-only to publish (bootstrap: publish, then a bounded wait for the other
//     workgroups), and again at the end with the lists live, so no row is lost and the expensive "early
//     phase" of a running top-k (every row beats an empty list) never happens.  (More than one max-only
//     tile -- B tiles put B x as many rows behind the first live threshold -- was built and measured in
//     round 4: candidates fall further, the time does not: the extra tiles cost what they save.)  Slots
//     only ever hold scores of real rows and only grow, so a stale read merely admits extra candidates:
//     speed, never correctness;
//   * at the end each lane appends its real entries to the query's compact candidate array
//     (one atomic add per lane) for k_select_* (select.hip);
//   * AP = true, the APPEND form (k <= 16 on 256-byte rows of a 16-bit scan, with the bootstrap): the shared
//     threshold alone decides what is kept, so a lane needs no sorted list -- a 4-entry unsorted register
//     buffer, flushed to the candidate array when full and at the end.  128 VGPRs instead of 223: TWO
//     workgroups per CU (four waves per SIMD) on twice the splits.
#include "scan.h"
#include "scan_dev.h"
#include <cstdlib>

#ifndef SSS_STAGGER
#define SSS_STAGGER 1
#endif

namespace sss {

// NW = waves per workgroup: 8 (two per SIMD, 256 VGPRs each) or, for 1024-byte rows whose resident
// queries alone take 128 VGPRs, 4 (one per SIMD, 512 VGPRs: no spills; NW * 32 queries per workgroup).
// THR = true is the THRESHOLD form (the rung between the fused search and the exhaustive kernels,
// ip_topk.hip: ip_topk_threshold): the queries are the compact list A.qsel, every lane compares against
// its query's FIXED threshold A.thr[] (scan domain) instead of a running list, and every row above it is
// appended to the query's candidate array -- no lists, no shared threshold, no bootstrap.
// waves whose bootstrap wait expired before the threshold existed (read + reset through scan_boot_expired)
__device__ unsigned g_boot_expired;

template <int RB, int TR, int DT, int NW, bool THR = false, bool AP = false>
__global__ __launch_bounds__(NW * 64, AP ? 2 * (NW / 4) : NW / 4) void k_scan(const ScanArgs A) {
    constexpr int H = TR / 64;                        // 64-row sub-steps per tile
    constexpr int CH = RB / 16;                       // 16-byte chunks per row
    constexpr int NU = RB / 32;                       // k-groups per row (one b128 fragment each)
    constexpr int TILE_BYTES = TR * RB;
    constexpr int LOADS_PER_WAVE = TR * CH / 64 / NW; // LDS-DMA wave-instructions per wave per tile
    constexpr int WGQ = NW * 32;                      // queries per workgroup
    constexpr bool PRECOMP = RB <= 512 && !AP;        // keep the DMA lane offsets in VGPRs (register budget; not at 128 VGPRs)
    constexpr int TAU_LDS = 2 * TILE_BYTES;           // [8 waves][32 queries][16 slots] u32 behind the two tile buffers
    static_assert(CH <= 64, "row longer than one LDS-DMA instruction");
    static_assert(LOADS_PER_WAVE >= 1, "a tile is at least one DMA piece per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nq = A.nq, n = A.n, S = A.S, G = A.G, J = A.J;
    const char* __restrict__ Qb = reinterpret_cast<const char*>(A.Q);
    const char* __restrict__ Cb = reinterpret_cast<const char*>(A.C);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware remap: blocks b and b+8 share an XCD (and its L2); the G query groups that
    // stream the same corpus split are given consecutive slots of ONE XCD so the split is
    // fetched from HBM once and re-read from that L2.  Speed only, never correctness.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int split = xcd * (S >> 3) + slot / G;
    const int g = slot % G;

    // ---- resident queries: lane (r, h) holds 16-byte chunk 2u + h of its query row in qc[u]
    const int q_local = wave * 32 + r;
    const int q_glob = g * WGQ + q_local;
    int q_ld = q_glob < nq ? q_glob : nq - 1;
    if constexpr (THR) q_ld = A.qsel[q_ld];             // row of Q this lane's (compact) query lives in
    f32x4 qc[NU];                                       // (loaded after the first tiles' DMA has been issued, below)

    // Lane list, sorted descending; empty slots are (-inf, -1).  thr = max(list tail, tau) is the
    // one value the hot path compares against.
    float ls[KP];
    int li[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) { ls[i] = -INFINITY; li[i] = -1; }
    float pend_s = -INFINITY;   // one parked candidate per lane (see the epilogue)
    int pend_i = -1;
    // AP (append form: K2 <= 16 with the bootstrap, 256-byte rows): the shared threshold alone decides what is kept,
    // so a lane needs no sorted list -- passing rows go into a small unsorted register buffer that is handed to the
    // query's candidate array when it is full and at the end (ls / li / pend_* are dead in this form).  Half the
    // registers: TWO workgroups per CU, four waves per SIMD.
    constexpr int E = 4;
    float es[E];
    int ei[E];
    int ecnt = 0;
#pragma unroll
    for (int i = 0; i < E; ++i) { es[i] = -INFINITY; ei[i] = -1; }
    float tau = -INFINITY, thr = -INFINITY;
    if constexpr (THR) thr = q_glob < nq ? A.thr[q_glob] : INFINITY;    // padding lanes never emit
    float rmax = -INFINITY;     // best score this lane has seen (published when cert == 1)
    float pub = -INFINITY;      // last value this lane published

    int tile_lo = split * A.tiles_per_split;
    int tile_hi = tile_lo + A.tiles_per_split;
    if (tile_hi > A.total_tiles) tile_hi = A.total_tiles;
    const int ntiles = tile_lo < tile_hi ? tile_hi - tile_lo : 0;
    const bool use_tau = !THR && J > 0;
    // bootstrap: the first tile of the split is scanned max-only first (iteration 0) and again, live, at the end
    // (iteration ntiles)
    const bool boot = use_tau && A.boot && ntiles > 0;
    const int nb = boot ? 1 : 0;
    const int niter = ntiles + nb;
    auto tile_of = [&](int i) { return i < ntiles ? tile_lo + i : tile_lo; };
    const int skip = A.tau_skip;                 // rank-selected threshold (cert == 1, J == 16): 16 - K2

    bool slots_seen = false;                     // the LDS copy of the slots has been filled at least once
    const unsigned* my_half = nullptr;
    unsigned cls_live = 0;                       // class of this lane's rows
    // cert == 1: a class is a set of SPLITS (split & 15; both lanes of a query's (h = 0, 1) pair belong to it), so the
    // pair publishes ONE value -- the better of its two lane maxima, by lane h = 0: half the agent-scope atomics (at the
    // bootstrap 128 instead of 256 per query line, which all arrive within a microsecond and serialise at the line's
    // memory channel).  cert > 1: a class per lane list, as the lists certify `cert` rows each.
#ifdef SSS_EXP_NOPAIR
    const bool pair_pub = false;
#else
    const bool pair_pub = A.cert == 1;
#endif
    if (use_tau) {
        // (split + split / 16: with the append form's 128 splits the low four bits of `split` alone would put the workgroups
        //  dispatched first -- the lower half of the grid, one per CU -- in classes 0-7 and their co-resident partners,
        //  which lose the SIMD arbitration and reach the end of the bootstrap tile ~10 us later, in classes 8-15: every
        //  wave then waits for the slow half before it has a threshold.  Mixed, each class has members of both halves.)
        cls_live = pair_pub ? (unsigned)(split + (split >> 4)) & 15u : (unsigned)(2 * split + h) % (unsigned)A.Ju;
        my_half = A.slots + (size_t)q_ld * SLOT_STRIDE + h * (J >> 1);
    }
    // Threshold word of the query from its J slots.  Synchronous form (bootstrap wait; J > 16): each lane of the
    // (h = 0, 1) pair reads half with agent-scope loads.  J == 16 with a rank: the (skip + 1)-th smallest; else the min.
    auto tau_ord = [&]() -> unsigned {
        if (J == 16 && skip > 0) {
            unsigned v[8];
            load8_sc1(my_half, v);
            return tau_select16(v, skip, h);
        }
        unsigned m = 0xFFFFFFFFu;
        for (int v = 0; v < (J >> 1); v += 8) m = min(m, min8_sc1(my_half + v));
        return min(m, (unsigned)__shfl_xor((int)m, 32));
    };
    // Asynchronous form (J == 16, every iteration): the wave's 32 queries x 64 B of slots are
    // fetched by two LDS-DMA instructions (agent scope) at the top of an iteration, land under the
    // MFMAs, are retired by the iteration's vmcnt(0) and read back by their owner lanes: no
    // stall, no registers held.
    const unsigned tau_lds = (unsigned)(unsigned long)(lptr_c)smem + TAU_LDS + wave * 2048;
    auto tau_fetch = [&]() {
        // (the lane offsets are recomputed from an OPAQUE copy of the lane id at every refresh: as loop invariants the
        //  compiler kept them in registers for the whole scan -- at the append form's register limit, in scratch, with a
        //  reload + vmcnt(0) in front of each of the two DMA instructions, which serialised them)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int qi = g * WGQ + wave * 32 + 16 * j + (ln >> 2);
            if (qi > nq - 1) qi = nq - 1;
            const unsigned off = (unsigned)qi * (unsigned)(SLOT_STRIDE * 4) + (unsigned)(ln & 3) * 16u;
            const unsigned dst = __builtin_amdgcn_readfirstlane(tau_lds + j * 1024);
            // (M0 is written without a save / restore: hipcc has no use of its own for M0 in this kernel -- no dynamic
            //  register indexing, no LDS-direct / GWS / sendmsg; every M0 reference in the ISA comes from these statements)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 sc1"
                         : : "v"(off), "s"(dst), "s"(A.slots) : "memory");
        }
    };
    auto tau_read = [&]() -> unsigned {
        const u32x4* p = reinterpret_cast<const u32x4*>(smem + TAU_LDS + wave * 2048 + r * 64 + h * 32);
        const u32x4 a = p[0], b = p[1];
        if (skip > 0) {
            unsigned v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            return tau_select16(v, skip, h);
        }
        const unsigned m = min(min(min(a.x, a.y), min(a.z, a.w)), min(min(b.x, b.y), min(b.z, b.w)));
        return min(m, (unsigned)__shfl_xor((int)m, 32));
    };
    auto set_tau = [&](unsigned m) {
        if (m > ORD_NEG_INF) tau = fmaxf(tau, ord2f(m - 1));     // the float just below the min slot
        thr = AP ? tau : fmaxf(ls[KP - 1], tau);
    };
    auto publish = [&](unsigned cls) {
        float val = rmax;
        bool ok = true;
        if (A.cert > 1) {
            // copies made opaque: a select chain over ls[] / li[] would be folded into a
            // runtime-indexed load and send both lists to scratch
            float v2 = ls[1], v4 = ls[3], v8 = ls[7], v16 = ls[KP - 1];
            int i2 = li[1], i4 = li[3], i8 = li[7], i16 = li[KP - 1];
            asm volatile("" : "+v"(v2), "+v"(v4), "+v"(v8), "+v"(v16), "+v"(i2), "+v"(i4), "+v"(i8), "+v"(i16));
            const int c = A.cert;
            val = c == 2 ? v2 : c == 4 ? v4 : c == 8 ? v8 : v16;
            ok = (c == 2 ? i2 : c == 4 ? i4 : c == 8 ? i8 : i16) >= 0;
        }
        // A lane's new best only matters if it beats its CLASS's best -- which ~10 lists share, so most lane records
        // are not class records.  The last fetched copy of the slots (LDS, J == 16) tells: without this filter the
        // early tiles, where every lane sets records all the time, spend most of their time waiting for some
        // hundred agent-scope atomics per query line to drain (vmcnt(0) at the end of the tile).
        if (pair_pub) { val = fmaxf(val, __shfl_xor(val, 32)); ok = h == 0; }
        unsigned cur = 0u;
        if (J == 16 && slots_seen) cur = *reinterpret_cast<const unsigned*>(smem + TAU_LDS + wave * 2048 + r * 64 + cls * 4u);
        if (ok && val > pub && val > tau && f2ord(val) > cur && q_glob < nq) {     // at or below tau it cannot raise the threshold
            // (the slot address is rebuilt from an opaque copy of the query index: a 64-bit per-lane pointer kept across the
            //  scan costs two registers the append form does not have -- it was spilled and reloaded here)
            int qq = THR ? q_ld : q_glob;
            asm volatile("" : "+v"(qq));
            __hip_atomic_fetch_max(A.slots + ((size_t)(unsigned)qq * (unsigned)SLOT_STRIDE + cls), f2ord(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pub = val;
        }
    };

    // LDS-DMA staging (global_load_lds_dwordx4, 1 KiB per wave-instruction).  Written as inline
    // asm so hipcc neither counts it nor drains vmcnt(0) at the next ds_read: the next tile
    // stays in flight under this tile's MFMAs and is retired by the explicit vmcnt(0) that
    // precedes the barrier at the end of the iteration (cdna_hip_programming.md section 5.7).
    // Slot p of the tile (16 B each) holds chunk (p % CH) ^ (row & 15) of row p / CH: the
    // swizzle is on the SOURCE address, the LDS image is lane-linear.
    const unsigned lds_base = (unsigned)(unsigned long)(lptr_c)smem;
    auto slot_row = [&](int i) { return ((wave * LOADS_PER_WAVE + i) * 64 + lane) / CH; };
    auto slot_off = [&](int i) {                    // byte offset of this lane's chunk inside the tile
        const int p = (wave * LOADS_PER_WAVE + i) * 64 + lane;
        const int tr = p / CH, sc = p % CH;
        return (unsigned)(tr * RB + ((sc ^ (tr & 15)) * 16));
    };
    unsigned lane_off[PRECOMP ? LOADS_PER_WAVE : 1];
    if (PRECOMP) {
#pragma unroll
        for (int i = 0; i < LOADS_PER_WAVE; ++i) lane_off[i] = slot_off(i);
    }
    // One LDS-DMA wave-instruction (piece i of this wave's share of a tile).
    auto stage_piece = [&](int buf, int tile_idx, int i) {
        const long row0 = (long)tile_idx * TR;
        const bool inside = row0 + TR <= (long)n;          // wave-uniform
        const char* tile_src = Cb + (size_t)row0 * RB;      // wave-uniform -> SGPR pair
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            lds_base + buf * TILE_BYTES + (wave * LOADS_PER_WAVE + i) * 1024);
        unsigned off = PRECOMP ? lane_off[PRECOMP ? i : 0] : slot_off(i);
        if (!inside) {                                       // ragged last tile: clamp the row
            const int lr = slot_row(i);
            long grow = row0 + lr;
            if (grow > (long)n - 1) grow = (long)n - 1;
            off = (unsigned)((grow - row0) * RB) + (off - (unsigned)(lr * RB));
        }
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                     : : "v"(off), "s"(dst), "s"(tile_src) : "memory");
    };
    auto stage = [&](int buf, int tile_idx) {
#pragma unroll
        for (int i = 0; i < LOADS_PER_WAVE; ++i) stage_piece(buf, tile_idx, i);
    };

    // per-lane LDS read swizzle term of chunk (2u + h) of row r
    const int x = h ^ (r & 15);
    f32x16 acc0 = {0}, acc1 = {0};

    auto mfma_sub = [&](int buf, int sub, int next_tile) {
        const char* tile = smem + buf * TILE_BYTES + sub * (64 * RB);
        // The next tile's DMA (all of this wave's pieces in one burst, once per tile): with a single
        // sub-step per tile it has to go out before this step's MFMAs to have time to land; otherwise
        // it follows the first sub-step's MFMAs, issuing while they execute, and has the rest of the
        // tile to land.  (One piece per k-group, as before, cost ~20 scalar instructions and a branch
        // per group in every step: the scan was bound by instruction issue, not by the matrix pipe.)
        if constexpr (H == 1) { if (next_tile >= 0) stage(buf ^ 1, next_tile); }
        auto lda = [&](int u) -> const f32x4* {
            const int c = (2 * u) ^ x;                          // == (2u + h) ^ (r & 15)
            return reinterpret_cast<const f32x4*>(tile + (r * CH + c) * 16);
        };
        // A fragments PF k-groups ahead of their MFMAs: the f32 MFMA spends 512 cycles on a group, one
        // group ahead covers the LDS latency; the 16-bit MFMAs spend 64-128, so their reads run further
        // ahead (all of a 256-byte row's fragments at once -- the registers are there).
        constexpr int PF = DT == DT_F32 ? 1 : AP ? 1 : (RB == 256 ? NU : RB == 512 ? 4 : 2);
        f32x4 as0[NU], as1[NU];
#pragma unroll
        for (int u = 0; u < PF && u < NU; ++u) { const f32x4* p = lda(u); as0[u] = p[0]; as1[u] = p[32 * CH]; }
        const f32x16 zero = {0};
        acc0 = zero; acc1 = zero;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + PF < NU) { const f32x4* p = lda(u + PF); as0[u + PF] = p[0]; as1[u + PF] = p[32 * CH]; }
            const f32x4 a0 = as0[u], a1 = as1[u];
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE this group's MFMAs
            if constexpr (DT == DT_F32) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, qc[u].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, qc[u].x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, qc[u].y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, qc[u].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, qc[u].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, qc[u].z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, qc[u].w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, qc[u].w, acc1, 0, 0, 0);
            } else if constexpr (DT == DT_BF16) {
                const bf16x8 qb = __builtin_bit_cast(bf16x8, qc[u]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), qb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), qb, acc1, 0, 0, 0);
            } else if constexpr (DT == DT_F16) {
                const f16x8 qb = __builtin_bit_cast(f16x8, qc[u]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), qb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), qb, acc1, 0, 0, 0);
            } else {
                // split f32: chunks of the hi half meet q_hi and q_lo, chunks of the lo half meet q_hi
                const bf16x8 A0 = __builtin_bit_cast(bf16x8, a0), A1 = __builtin_bit_cast(bf16x8, a1);
                const bf16x8 qh = __builtin_bit_cast(bf16x8, qc[u < NU / 2 ? u : u - NU / 2]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, qh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, qh, acc1, 0, 0, 0);
                if (u < NU / 2) {
                    const bf16x8 ql = __builtin_bit_cast(bf16x8, qc[u + NU / 2]);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, ql, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, ql, acc1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (H > 1) { if (sub == 0 && next_tile >= 0) stage(buf ^ 1, next_tile); }
    };

    auto block_max = [&](const f32x16& a, float& q0, float& q1, float& q2, float& q3) {
        q0 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
        q1 = fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7]));
        q2 = fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11]));
        q3 = fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]));
        return fmaxf(fmaxf(q0, q1), fmaxf(q2, q3));
    };
    // Rare path of the top-k epilogue of one 32x32 accumulator: a[j] is (corpus row base + (j&3) +
    // 8*(j>>2), query r).  Only the quarters (rows 8g..8g+3 of this lane's 16) that hold a passing
    // score are walked.  A passing score parks in the lane's one pending slot; the 80-instruction
    // sorted insert runs only when some lane needs its slot again (then every lane's pending entry
    // goes in with that same pass).  thr may therefore lag behind -- it only admits extra
    // candidates, never drops one.
    auto insert_block = [&](const f32x16& a, int base) {
        float q0, q1, q2, q3;
        const float m = block_max(a, q0, q1, q2, q3);
        if (__builtin_amdgcn_ballot_w64(m > thr) == 0) return;
        auto walk = [&](float qm, int j0) {
            if (__builtin_amdgcn_ballot_w64(qm > thr) == 0) return;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj;
                const bool pass = a[j] > thr;
                if (__builtin_amdgcn_ballot_w64(pass) != 0) {
                    if (__builtin_amdgcn_ballot_w64(pass && pend_i >= 0) != 0) {
                        list_insert<KP>(ls, li, pend_s, pend_i);
                        pend_s = -INFINITY; pend_i = -1;
                        thr = fmaxf(ls[KP - 1], tau);
                    }
                    const bool still = a[j] > thr;
                    pend_s = still ? a[j] : pend_s;
                    pend_i = still ? base + (j & 3) + 8 * (j >> 2) : pend_i;
                }
            }
        };
        walk(q0, 0); walk(q1, 4); walk(q2, 8); walk(q3, 12);     // ascending row order per lane
    };
    // AP: lanes flagged `need` hand their buffered rows to the query's candidate array (one atomic add per lane; what
    // does not fit the capacity is remembered as the largest lost key, exactly like the tail of a full list).
    auto flush = [&](bool need) {
        if (need && q_glob < nq) {
            const unsigned at = atomicAdd(A.cnt + q_glob, (unsigned)ecnt);
            unsigned long long* dst = A.cand + (size_t)q_glob * A.cap;
            unsigned long long lost = 0ull;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                if (i < ecnt) {
                    const unsigned long long key = make_key(es[i], ei[i]);
                    if (at + (unsigned)i < (unsigned)A.cap) dst[at + i] = key;
                    else lost = lost > key ? lost : key;
                }
            }
            if (lost != 0ull && A.maxlast != nullptr) atomicMax(A.maxlast + q_glob, lost);   // (threshold form: the count alone tells)
        }
        ecnt = need ? 0 : ecnt;
    };
    auto append_block = [&](const f32x16& a, int base) {
        float q0, q1, q2, q3;
        const float m = block_max(a, q0, q1, q2, q3);
        if (__builtin_amdgcn_ballot_w64(m > thr) == 0) return;
        auto walk = [&](float qm, int j0) {
            if (__builtin_amdgcn_ballot_w64(qm > thr) == 0) return;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj;
                const bool pass = a[j] > thr;
                if (__builtin_amdgcn_ballot_w64(pass) != 0) {
                    const bool full = pass && ecnt == E;
                    if (__builtin_amdgcn_ballot_w64(full) != 0) flush(full);
#pragma unroll
                    for (int i = E - 1; i > 0; --i) { es[i] = pass ? es[i - 1] : es[i]; ei[i] = pass ? ei[i - 1] : ei[i]; }
                    es[0] = pass ? a[j] : es[0];
                    ei[0] = pass ? base + (j & 3) + 8 * (j >> 2) : ei[0];
                    ecnt += pass ? 1 : 0;
                }
            }
        };
        walk(q0, 0); walk(q1, 4); walk(q2, 8); walk(q3, 12);
    };
    // THR: every score above the lane's fixed threshold goes straight to the query's candidate array
    // (rare by construction: the threshold sits an error bound below the k-th best score already known).
    auto emit_block = [&](const f32x16& a, int base) {
        float q0, q1, q2, q3;
        const float m = block_max(a, q0, q1, q2, q3);
        if (__builtin_amdgcn_ballot_w64(m > thr) == 0) return;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (a[j] > thr) {
                const unsigned pos = atomicAdd(A.cnt + q_glob, 1u);
                if (pos < (unsigned)A.cap) A.cand[(size_t)q_glob * A.cap + pos] = make_key(a[j], base + (j & 3) + 8 * (j >> 2));
            }
        }
    };
    // Prologue: the first TWO tiles' DMA goes out before anything else (both buffers are free; from a cold start a tile
    // takes ~4 us to land, far longer than the bootstrap tile takes to scan), then the query rows are fetched and
    // converted under it.
    const bool two_ahead = H > 1 && niter > 1;          // (H == 1 issues a tile's successor before its MFMAs anyway)
    if (ntiles > 0) stage(0, tile_lo);
    if (two_ahead) stage(1, tile_of(1));
    load_queries<RB, DT>(Qb, q_ld, h, qc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // The scan advances in 64-row steps, t = tile iteration * H + sub-step.  The loop is split in two
    // so that the HOT loop (MFMAs, one max tree and one compare per step) never writes the list
    // state: a step whose maximum beats some lane's threshold leaves the hot loop, runs the rare
    // insert path on the still-live accumulators and re-enters.  (Written as one loop with the rare
    // path inside, the register allocator cannot keep the 35 state registers in place and pays
    // ~70 v_mov per step on the hot path.)
    const int T = niter * H;
    long row0_of_step = 0;                      // first corpus row of the current step
    // The threshold moves fast at first and then ever more slowly, and a stale one only admits extra candidates.  A
    // refresh is not cheap: its two LDS-DMA instructions (and the publish's atomic) queue behind the tile traffic of the
    // CU's memory pipe -- ~3 k cycles per wave, half a 128-row tile of a 16-bit scan (per-tile stamps).
    // With tau near the R-th best of the i tiles' rows seen so far, ~R / i rows per query pass per tile-time and a
    // threshold stale by D tiles admits ~R D / i^2 more: the cost of refreshing every D tiles, c_r / D + c_p D / i^2
    // per tile, is least at D ~ i -- so every scan refreshes at i = 2, 3, 4 and then at 2 and 3 times the powers of
    // two (6, 8, 12, 16, 24, ...: 12 refreshes of a 61-tile split instead of 21, 18 of 610 instead of 158).  The f32
    // scan (15 us per tile) refreshed every tile until round 4; the same schedule takes 2 % off it (4 % at 125 k rows).
    constexpr bool TAU_EVERY_TILE = TR >= 512;
    auto refresh_at = [&](int i) {
        // (no refresh in the tile right behind the bootstrap: the wave has just polled its threshold, and the first DMA
        //  fetch of the slot lines -- while every CU's bootstrap atomics are still draining at the memory side -- took
        //  ~25 k cycles to issue: per-tile stamps, round 4)
        if (i == 1 && boot && !TAU_EVERY_TILE) return false;
        if (TAU_EVERY_TILE || i <= 4) return true;
        const int sh = 30 - __builtin_clz(i);              // i = (2 or 3) << sh  <=>  its low sh bits are zero
        return (i & ((1 << sh) - 1)) == 0;
    };
    auto tile_top = [&](int i) {                // threshold refresh at the start of tile iteration i > 0
        if (!use_tau || i == 0) return;
        if (J == 16) {
            if (!refresh_at(i)) return;
            publish(cls_live);                  // (before the fetch: an atomic behind the two DMA instructions waits for them)
            tau_fetch();                        // lands under this tile's MFMAs
            return;
        }
        // J > 16 (k > 116): synchronous loads, rarely, staggered between the two waves of a
        // SIMD (waves 4-7 one tile later) so the partner keeps the matrix pipe busy meanwhile.
        const int ii = i - (wave >= 4 ? 1 : 0);        // (NW == 4: no SIMD partner, nothing to stagger)
        if (ii >= 2 && (ii <= 8 || (ii & (ii - 1)) == 0 || (ii & 15) == 0)) set_tau(tau_ord());
        publish(cls_live);                      // completes under this tile's MFMAs
    };
    auto tile_end = [&](int i) {
        const bool pre = boot && i == 0;
        if (pre) publish(cls_live);
        // this wave's share of the next tile landed (after the bootstrap tile: its own atomics did -- skipping this wait
        // there only moves it into the polling loop below, measured slower)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (use_tau && J == 16 && i > 0 && refresh_at(i)) { set_tau(tau_read()); slots_seen = true; }   // ... and so did its slots (own region: no barrier needed)
        __syncthreads();                                   // ... everyone's did, and this buffer is free
        if (pre) {
            // Wait (bounded) until every class of this wave's queries has published its bootstrap
            // maximum.  All workgroups of a launch are normally co-resident and reach this point
            // within a microsecond of each other; if not, the bound expires and the scan simply
            // runs with a weaker (or no) threshold -- correctness never depends on it.
            unsigned m = 0;
            // (the append form has nothing but the threshold to hold rows back: it waits much longer before it gives up)
            for (int it = 0; it < (AP ? 1024 : 24); ++it) {
                m = tau_ord();
                if (__builtin_amdgcn_ballot_w64(m == 0) == 0) break;
                __builtin_amdgcn_s_sleep(16);
            }
            // (debug counter, sss_scan_boot_expired: the wait ran out before K2 classes of every query had published --
            //  the workgroups of the launch were not co-resident; correct all the same, but the append form then floods)
            if (lane == 0 && __builtin_amdgcn_ballot_w64(m == 0 && q_glob < nq) != 0) atomicAdd(&g_boot_expired, 1u);
            set_tau(m);
        }
    };
    // MFMAs of step t into acc0 / acc1 (acc[j] is (corpus row row0 + (j&3) + 8*(j>>2) + 4h, query r),
    // acc1 32 rows further); score_tree() returns the lane's maximum over both.
    auto score_mfma = [&](int t) {
        const int i = (int)((unsigned)t / (unsigned)H), sub = (int)((unsigned)t % (unsigned)H);   // (unsigned: shifts, not the signed-division sequence)
        if (sub == 0) tile_top(i);
        const int next_tile = (i + 1 < niter && !(two_ahead && i == 0)) ? tile_of(i + 1) : -1;
        mfma_sub(i & 1, sub, next_tile);
        row0_of_step = (long)tile_of(i) * TR + sub * 64;
        if (row0_of_step + 64 > n) {                        // wave-uniform, last tile only
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rr = (int)row0_of_step + 4 * h + (j & 3) + 8 * (j >> 2);
                if (rr >= n) acc0[j] = -INFINITY;
                if (rr + 32 >= n) acc1[j] = -INFINITY;
            }
        }
    };
    auto score_tree = [&]() -> float {
        // 16 v_max3 (as asm: fmaxf() adds a canonicalising v_max x, x per MFMA result it touches)
        float m0 = vmax3(acc0[0], acc0[1], acc0[2]), m1 = vmax3(acc1[0], acc1[1], acc1[2]);
#pragma unroll
        for (int j = 3; j < 15; j += 2) { m0 = vmax3(m0, acc0[j], acc0[j + 1]); m1 = vmax3(m1, acc1[j], acc1[j + 1]); }
        const float m = vmax3(m0, m1, acc0[15]);
        const float mm = vmax3(m, acc1[15], rmax);
        rmax = mm;
        return vmax3(m, acc1[15], acc1[15]);
    };
    const int t_live = nb * H;                  // steps of the bootstrap tile: lane maximum only
    // STAGGER (MI355X_MICROARCH.md, two waves per SIMD, item 9): the two waves of a SIMD run the same program and
    // leave every barrier together -- both into their MFMAs, then both into their max trees, the matrix pipe idle
    // meanwhile.  Waves 4-7 (the SIMD partners of waves 0-3) therefore take the end of a tile -- wait, threshold
    // refresh, barrier -- BEFORE the epilogue of its last step instead of after it (the accumulators simply stay
    // live across the barrier): after every barrier one partner starts with matrix work, the other with vector work.
    // Measured (same device, alternating builds): split scan -3.5 %, bf16 C5 -1.3 % time; f16 and f32 scans unchanged
    // to +1 % (their partners drift apart by themselves), so those keep the plain order.
    const bool defer = SSS_STAGGER && NW == 8 && (DT == DT_SPLIT || DT == DT_BF16) && wave >= 4;
    int t = 0;
    if constexpr (AP || THR) {
        // The forms WITHOUT lane lists (append, threshold) have no list state to keep out of the hot loop, so theirs is the
        // plain nest: tiles x (compile-time) sub-steps, the rare path an ordinary side branch.  What that buys is SCALAR
        // instructions: a SIMD issues one scalar instruction per 4 cycles, and the split loop below spends ~90 per
        // 64-row step on t / H, t % H, tile-of-iteration, 64-bit row arithmetic and the ragged-tile test -- with four
        // waves per SIMD that is ~70 % of the scalar issue slots of a step, the append form's real limit (the matrix
        // pipe is ~70 % busy, the two co-resident workgroups together finish in the same time however the SIMDs
        // arbitrate between them: section 5.1).  Here the per-tile values are computed once per tile.
        t = T;                                                  // (the split loop below is not entered)
        for (int i = 0; i < niter; ++i) {
            tile_top(i);
            const int tile = tile_of(i);
            const int next_tile = (i + 1 < niter && !(two_ahead && i == 0)) ? tile_of(i + 1) : -1;
            const int row_base = tile * TR;
            const bool ragged = row_base + TR > n;              // wave-uniform, last tile of the corpus only
            const bool live = i >= nb;
            if constexpr (THR) {
                // a wave whose 32 query slots are all padding (the rung's compact query list rarely fills the 256 slots of
                // a workgroup: ~100 unproven queries of a 1024-query batch leave waves 4-7 empty) stages its share of the
                // tiles and keeps the barriers, nothing else: the three-pass split scan of the rung is bound by the matrix
                // pipe, and an empty wave was taking half of its SIMD's
                if (g * WGQ + wave * 32 >= nq) {
                    if (next_tile >= 0) stage((i & 1) ^ 1, next_tile);
                    tile_end(i);
                    continue;
                }
            }
            // (unrolled: a single copy of the sub-step -- `#pragma unroll 1`, 6.4 k lines of ISA instead of 11.3 k, the
            //  rare path's ~20 KB of code once instead of twice -- measured the same at 1M rows and 5 % SLOWER at 10M)
#pragma unroll
            for (int sub = 0; sub < H; ++sub) {
                mfma_sub(i & 1, sub, next_tile);
                const int row0 = row_base + sub * 64;
                if (ragged) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int rr = row0 + 4 * h + (j & 3) + 8 * (j >> 2);
                        if (rr >= n) acc0[j] = -INFINITY;
                        if (rr + 32 >= n) acc1[j] = -INFINITY;
                    }
                }
                const float m = score_tree();
                if (live && __builtin_amdgcn_ballot_w64(m > thr) != 0) {
                    // (the threshold form used to hand every passing row to the candidate array with an atomic add of its own
                    //  -- fine for the rung's few queries and tight thresholds; the sampled large-k search keeps thousands of
                    //  rows per query: the append form's 4-entry register buffer, one atomic add per flush, serves both)
                    append_block(acc0, row0 + 4 * h);
                    append_block(acc1, row0 + 32 + 4 * h);
                }
            }
            tile_end(i);
        }
    }
    while (t < T) {
        bool rare = false;
        for (; t < T; ++t) {                    // ---- hot loop
            const bool last = (unsigned)t % (unsigned)H == H - 1;
            const bool early = defer && t >= t_live;        // (not on the bootstrap tile: its end publishes the tile's maxima)
            score_mfma(t);
            if (early && last) tile_end((int)((unsigned)t / (unsigned)H));
            const float m = score_tree();
            if (t >= t_live && __builtin_amdgcn_ballot_w64(m > thr) != 0) { rare = true; break; }
            if (!early && last) tile_end((int)((unsigned)t / (unsigned)H));
        }
        if (!rare) break;
        if constexpr (THR) {
            emit_block(acc0, (int)row0_of_step + 4 * h);
            emit_block(acc1, (int)row0_of_step + 32 + 4 * h);
        } else if constexpr (AP) {
            append_block(acc0, (int)row0_of_step + 4 * h);
            append_block(acc1, (int)row0_of_step + 32 + 4 * h);
        } else {
            insert_block(acc0, (int)row0_of_step + 4 * h);
            insert_block(acc1, (int)row0_of_step + 32 + 4 * h);
        }
        if (!defer && (unsigned)t % (unsigned)H == H - 1) tile_end((int)((unsigned)t / (unsigned)H));    // (rare implies t >= t_live)
        ++t;
    }
    if constexpr (THR) { flush(ecnt > 0); return; }
    if constexpr (AP) flush(ecnt > 0);
    list_insert<KP>(ls, li, pend_s, pend_i);   // no-op for lanes with an empty slot (-inf)
    // ---- append the real entries to the query's compact candidate array (none in the append form: li stayed -1)
    if (!AP && q_glob < nq) {
        int nreal = 0;
#pragma unroll
        for (int i = 0; i < KP; ++i) nreal += li[i] >= 0 ? 1 : 0;
        if (nreal > 0) {
            const unsigned base = atomicAdd(A.cnt + q_glob, (unsigned)nreal);
            unsigned long long* dst = A.cand + (size_t)q_glob * A.cap + base;
#pragma unroll
            for (int i = 0; i < KP; ++i)
                if (i < nreal) dst[i] = make_key(ls[i], li[i]);
            if (nreal == KP) atomicMax(A.maxlast + q_glob, (unsigned long long)make_key(ls[KP - 1], li[KP - 1]));
        }
    }
}

// ------------------------------------------------------------------------------ host side
// Number of waves, since the last reset, whose bootstrap wait ran out (synchronises the device: a debugging / bench aid).
int scan_boot_expired(int reset) {
    unsigned v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_boot_expired), sizeof(v)) != hipSuccess) { set_error("scan_boot_expired: read failed"); return SSS_EHIP; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_boot_expired), &zero, sizeof(zero)) != hipSuccess) { set_error("scan_boot_expired: reset failed"); return SSS_EHIP; }
    return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);
}

int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev < 0 || dev >= MAX_DEVICES ? 0 : dev;
}

static int pick_splits(long n, int G, int tr) {
    // S*G workgroups, one per CU (256 CUs); S a multiple of 8 (XCD remap); >= one tile a split.
    int S = (256 / G) & ~7;
    if (S < 8) S = 8;
    while (S > 8 && (long)S * tr > n) S -= 8;
    return S;
}

constexpr int TR256_MIN_TILES = 24;     // splits at least this many 256-row tiles long use them (256-byte rows)

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static ScanPlan make_plan_for(long nq, long n, int d, int k, int dtype, bool want_append);

// The append form's bootstrap hand-shake needs all 2 x 256 workgroups resident at once: ask the runtime, once per
// device, whether two of its workgroups (80 KB of LDS, 128 VGPRs each) fit a CU and the chip has the CUs -- otherwise
// (another LDS carve-out, a smaller part) the plan keeps the list form.
static bool append_form_fits() {
    static int cached[MAX_DEVICES] = {};                 // 0 unknown, 1 yes, 2 no
    const int dev = current_device();
    if (cached[dev] == 0) {
        constexpr int lds = 2 * 128 * 256 + 8 * 2048;
        const void* fn = reinterpret_cast<const void*>(&k_scan<256, 128, DT_F16, 8, false, true>);
        int blocks = 0, cus = 0;
        bool ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 512, lds) == hipSuccess &&
                  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
        (void)hipGetLastError();
        cached[dev] = (ok && blocks >= 2 && cus >= 256) ? 1 : 2;
    }
    return cached[dev] == 1;
}

ScanPlan make_plan(long nq, long n, int d, int k, int dtype) {
    // The append form (k <= 16 on 256-byte rows of a 16-bit scan: no lane lists, half the registers) runs TWO
    // workgroups per CU on twice the splits; it needs the bootstrap (a shared threshold from the first live row on) and
    // splits long enough to be worth it -- otherwise the plan with lane lists.
    static const int ap_off = getenv("SSS_SCAN_AP_OFF") ? 1 : 0;       // (dev A/B switch)
    const int rb = d * elem_bytes(dtype);
    if (!ap_off && rb == 256 && dtype != DT_F32 && k <= KP && append_form_fits()) {
        const ScanPlan a = make_plan_for(nq, n, d, k, dtype, true);
        if (a.append) return a;
    }
    return make_plan_for(nq, n, d, k, dtype, false);
}
static ScanPlan make_plan_for(long nq, long n, int d, int k, int dtype, bool want_append) {
    ScanPlan p;
    const int rb = d * elem_bytes(dtype);
    const int wgq = rb == 1024 ? 128 : WG_QUERIES;      // queries per workgroup (k_scan's NW * 32)
    p.G = (int)((nq + wgq - 1) / wgq);
    // 128-row tiles (one barrier per 128 rows) for long splits; 64-row tiles keep the split
    // granularity (and the once-repeated bootstrap tile) small when a split is only a few tiles.
    int tr = rb <= 512 ? 128 : 64;
    int S = pick_splits(n, p.G, tr);
    if (rb == 512 && (n + (long)S * 128 - 1) / ((long)S * 128) < 48) { tr = 64; S = pick_splits(n, p.G, tr); }
    // 256-byte rows: 256-row tiles (one barrier and one threshold refresh per 256 rows, the next tile's
    // DMA a whole tile ahead) once a split is long enough to amortise the twice-scanned bootstrap tile
    if (rb == 256 && dtype != DT_F32 && n / ((long)S * 256) >= TR256_MIN_TILES) tr = 256;
    if (want_append) {                                    // 2 S splits of 128-row tiles, S * G <= 256
        tr = 128;
        S = pick_splits(n, p.G, tr);
        // (>= 7 tiles a split after doubling: measured round 4 -- 125 k rows, one of eight shards of the 1M corpus, 0.122 ->
        //  0.110 ms with the append form; 100 k rows the same either way, below that the list form)
        if ((long)S * p.G > 256 || (long)2 * S * tr * 7 > n) want_append = false;
        else S *= 2;
        if (!want_append) { ScanPlan none; none.append = 0; return none; }
    }
    p.tile_rows = tr;
    p.S = S;
    p.L = 2 * S;
    // k <= 14: K2 = max(k + 2, 8) (class maxima + bootstrap; the select kernel's second chance widens the candidate
    // set where the slack K2 - k is too thin for a query -- as it always had to for k = 15, 16, where K2 = 16 = KP
    // leaves none); larger k: k + 12.  A small K2 pays twice: the shared threshold certifies only K2 rows, and
    // "min over K2 class maxima" sits nearer the top the fewer classes there are (it tracks roughly the 35th best
    // row seen with 12 classes, the 50th with 16): a third fewer candidates to insert, re-score and carry.
    p.K2 = k + 2 <= KP ? (k + 2 < 8 ? 8 : k + 2) : k <= KP ? KP : k + 12;
    if ((long)p.L * KP < p.K2) p.K2 = p.L * KP;
    p.total_tiles = (int)((n + tr - 1) / tr);
    p.tiles_per_split = (p.total_tiles + S - 1) / S;
    p.cap = p.L * KP;
    // threshold slots: J classes x cert entries certify J * cert >= K2 rows (scan.hip header).
    // K2 <= 16: 16 classes of class maxima (cert 1, with the bootstrap; refreshed through LDS-DMA
    // every iteration).  Larger K2: 16 (64 beyond K2 = 128) classes certify `cert` rows each with
    // their lists' cert-th best -- measured faster than one class per row, whose 7+ slot loads per
    // refresh have to be synchronous.
    const int active_splits = (p.total_tiles + p.tiles_per_split - 1) / p.tiles_per_split;
    p.J = p.K2 <= 128 ? 16 : 64;
    p.cert = 1;
    while (p.J * p.cert < p.K2 && p.cert < KP) p.cert *= 2;
    // cert == 1 (K2 <= 16): all 16 slots are distinct classes and the threshold is the K2-th largest class maximum
    // (tau_skip = 16 - K2 words allowed below it; scan_dev.h: tau_select16)
    p.Ju = p.J;
    p.tau_skip = (p.cert == 1 && p.J == 16) ? 16 - p.K2 : 0;
    if (p.J * p.cert < p.K2 || 2 * active_splits < p.Ju) { p.J = 0; p.Ju = 0; p.tau_skip = 0; }      // tiny corpus: no threshold
    p.boot = (p.J > 0 && p.cert == 1) ? 1 : 0;
    p.append = (want_append && p.boot) ? 1 : 0;
    if (want_append && !p.append) return p;               // (the caller falls back to the list plan)
    if (p.append) p.cap = 2048;                            // candidates per query the select kernels stage (FS_CAP)
    p.total_bytes = align256((size_t)nq * p.cap * 8);
    return p;
}

// Plan of the threshold form: G groups of compact queries x S corpus splits, 128-row tiles (64 for
// 1024-byte rows), no threshold slots; cap = candidates kept per query.
ScanPlan make_thr_plan(long nsel, long n, int d, int scan_dtype, int cap) {
    ScanPlan p = {};
    const int rb = d * elem_bytes(scan_dtype);
    const int wgq = rb == 1024 ? 128 : WG_QUERIES;
    p.G = (int)((nsel + wgq - 1) / wgq);
    p.tile_rows = rb <= 512 ? 128 : 64;
    p.S = pick_splits(n, p.G, p.tile_rows);
    p.L = 2 * p.S;
    p.total_tiles = (int)((n + p.tile_rows - 1) / p.tile_rows);
    p.tiles_per_split = (p.total_tiles + p.S - 1) / p.S;
    p.cap = cap;
    p.total_bytes = align256((size_t)nsel * cap * 8);
    return p;
}

template <int RB, int TR, int DT, int NW, bool THR, bool AP = false>
static int launch_form(const ScanArgs& a, hipStream_t st) {
    const size_t lds = 2 * (size_t)TR * RB + NW * 2048;      // two tile buffers + the threshold-slot staging
    static bool attr_done[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr_done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan<RB, TR, DT, NW, THR, AP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((k_scan<RB, TR, DT, NW, THR, AP>), dim3(a.S * a.G), dim3(NW * 64), lds, st, a);
    return check_launch("k_scan");
}
template <int RB, int TR, int DT, int NW = 8>
static int launch_one(const ScanArgs& a, hipStream_t st) {
    // the threshold form is compiled for one tile shape per row size (make_thr_plan picks it)
    if (a.thr != nullptr) {
        if constexpr (TR == (RB <= 512 ? 128 : 64)) return launch_form<RB, TR, DT, NW, true>(a, st);
        set_error("scan: threshold form not built for %d-row tiles of %d-byte rows", TR, RB);
        return SSS_EINVAL;
    }
    if (a.append) {
        if constexpr (RB == 256 && TR == 128 && DT != DT_F32) return launch_form<RB, TR, DT, NW, false, true>(a, st);
        set_error("scan: append form not built for this shape");
        return SSS_EINVAL;
    }
    return launch_form<RB, TR, DT, NW, false>(a, st);
}

int launch_scan(int dtype, int d, int tile_rows, const ScanArgs& a, hipStream_t st) {
    const int rb = d * elem_bytes(dtype);
    if (dtype == DT_F32) {
        if (rb == 256) return tile_rows == 256 ? launch_one<256, 256, DT_F32>(a, st) : launch_one<256, 128, DT_F32>(a, st);
        if (rb == 512) return tile_rows == 128 ? launch_one<512, 128, DT_F32>(a, st) : launch_one<512, 64, DT_F32>(a, st);
        if (rb == 1024) return launch_one<1024, 64, DT_F32, 4>(a, st);
    } else if (dtype == DT_BF16) {
        if (rb == 256) return tile_rows == 256 ? launch_one<256, 256, DT_BF16>(a, st) : launch_one<256, 128, DT_BF16>(a, st);
        if (rb == 512) return tile_rows == 128 ? launch_one<512, 128, DT_BF16>(a, st) : launch_one<512, 64, DT_BF16>(a, st);
        if (rb == 1024) return launch_one<1024, 64, DT_BF16, 4>(a, st);
    } else if (dtype == DT_F16) {
        if (rb == 256) return tile_rows == 256 ? launch_one<256, 256, DT_F16>(a, st) : launch_one<256, 128, DT_F16>(a, st);
        if (rb == 512) return tile_rows == 128 ? launch_one<512, 128, DT_F16>(a, st) : launch_one<512, 64, DT_F16>(a, st);
        if (rb == 1024) return launch_one<1024, 64, DT_F16, 4>(a, st);
    } else if (dtype == DT_SPLIT) {
        if (rb == 256) return tile_rows == 256 ? launch_one<256, 256, DT_SPLIT>(a, st) : launch_one<256, 128, DT_SPLIT>(a, st);
        if (rb == 512) return tile_rows == 128 ? launch_one<512, 128, DT_SPLIT>(a, st) : launch_one<512, 64, DT_SPLIT>(a, st);
        if (rb == 1024) return launch_one<1024, 64, DT_SPLIT, 4>(a, st);
    }
    set_error("scan: unsupported row size %d bytes", rb);
    return SSS_EINVAL;
}

}  // namespace sss
