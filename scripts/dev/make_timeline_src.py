"""Dev helper (not product): writes csrc/scan_tl.hip = csrc/scan.hip + in-kernel stamps (s_memrealtime / s_memtime),
a rare-path counter and the symbol sss_debug_timeline.  Anchored on source lines, so it follows the kernel as it
changes (the old timeline.patch did not).  Build:
    python scripts/dev/make_timeline_src.py && cd sessionsimilaritysearch_amd/csrc && make &&
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c scan_tl.hip -o scan_tl.o &&
    hipcc --offload-arch=gfx950 -shared -o ../../scripts/dev/libsss_tl.so scan_tl.o $(ls *.o | grep -v '^scan\\.o$' | grep -v scan_tl.o)
Slots per workgroup (16 x u64): 0 realtime start, 1..5 memtime (start, queries loaded, first tile landed, loop end,
kernel end), 6 rare-path entries, 7 realtime end, 8 memtime after the bootstrap wait, 9 realtime at loop start, 10 cycles wave 0 spent inside the rare path."""
import os, re
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "sessionsimilaritysearch_amd/csrc/scan.hip")).read()

def sub1(s, old, new):
    assert s.count(old) == 1, (s.count(old), old)
    return s.replace(old, new)

src = sub1(src, "template <int RB, int TR, int DT, int NW, bool THR = false, bool AP = false>\n",
    "__device__ unsigned long long g_tl[1024 * 16];\n__device__ unsigned long long g_tiles[192];\n"
    "#define TL(slot) do { if (threadIdx.x == 0) g_tl[blockIdx.x * 16 + (slot)] = ((slot) == 0 || (slot) == 7 || (slot) == 9) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)\n"
    "template <int RB, int TR, int DT, int NW, bool THR = false, bool AP = false>\n")
src = sub1(src, "    constexpr int H = TR / 64;", "    TL(0); TL(1);\n    constexpr int H = TR / 64;")
src = sub1(src, "    if (ntiles > 0) stage(0, tile_lo);\n", "    TL(2);\n    if (ntiles > 0) stage(0, tile_lo);\n")
src = sub1(src, "    __syncthreads();\n\n    // The scan advances in 64-row steps", "    __syncthreads();\n    TL(3); TL(9);\n\n    // The scan advances in 64-row steps")
src = sub1(src, "            set_tau(m);\n        }\n    };", "            set_tau(m);\n            TL(8);\n        }\n        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 101) && i < 96) g_tiles[(blockIdx.x ? 96 : 0) + i] = __builtin_amdgcn_s_memtime();\n    };")
src = sub1(src, "    int t = 0;\n    if constexpr (AP || THR) {", "    int t = 0;\n    int n_rare = 0;\n    unsigned long long rare_cyc = 0;\n    if constexpr (AP || THR) {")
# the list-free forms' nested loop: rare-path entries and cycles
src = sub1(src, "                if (live && __builtin_amdgcn_ballot_w64(m > thr) != 0) {\n",
    "                if (live && __builtin_amdgcn_ballot_w64(m > thr) != 0) {\n                    ++n_rare;\n                    const unsigned long long rcn0 = __builtin_amdgcn_s_memtime();\n")
src = sub1(src, "                        append_block(acc1, row0 + 32 + 4 * h);\n                    }\n",
    "                        append_block(acc1, row0 + 32 + 4 * h);\n                    }\n                    rare_cyc += __builtin_amdgcn_s_memtime() - rcn0;\n")
src = sub1(src, "        if (!rare) break;\n", "        if (!rare) break;\n        ++n_rare;\n        const unsigned long long rc0 = __builtin_amdgcn_s_memtime();\n")
src = sub1(src, "        if (!defer && (unsigned)t % (unsigned)H == H - 1) tile_end((int)((unsigned)t / (unsigned)H));    // (rare implies t >= t_live)\n",
    "        rare_cyc += __builtin_amdgcn_s_memtime() - rc0;   // (NB: hipcc may hoist the next step's MFMAs above this stamp when no tile ends here)\n        if (!defer && (unsigned)t % (unsigned)H == H - 1) tile_end((int)((unsigned)t / (unsigned)H));\n")
src = sub1(src, "    if constexpr (THR) return;\n",
    "    TL(4);\n    if (threadIdx.x == 0) { g_tl[blockIdx.x * 16 + 6] = (unsigned long long)n_rare; g_tl[blockIdx.x * 16 + 10] = rare_cyc; }\n"
    "    if constexpr (THR) return;\n")
# kernel end: the closing brace right before the host-side banner
src = sub1(src, "        }\n    }\n}\n\n// ------------------------------------------------------------------------------ host side",
    "        }\n    }\n    TL(5); TL(7);\n}\n"
    "extern \"C\" int sss_debug_timeline(unsigned long long* host, int n) {\n"
    "    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sss::g_tl), (size_t)n * 8);\n}\n"
    "extern \"C\" int sss_debug_tiles(unsigned long long* host) {\n"
    "    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sss::g_tiles), 192 * 8);\n}\n"
    "\n// ------------------------------------------------------------------------------ host side")
src = sub1(src, "constexpr int TR256_MIN_TILES = 24;", "static const int TR256_MIN_TILES = getenv(\"SSS_TR256_MIN_TILES\") ? atoi(getenv(\"SSS_TR256_MIN_TILES\")) : 24;")
src = sub1(src, '#include "scan.h"\n', '#include "scan.h"\n#include <cstdlib>\n')
src = sub1(src, "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n",
    "        const unsigned long long w0 = __builtin_amdgcn_s_memtime();\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        const unsigned long long w1 = __builtin_amdgcn_s_memtime();\n")
src = sub1(src, "        __syncthreads();                                   // ... everyone's did, and this buffer is free\n",
    "        __syncthreads();\n        if (threadIdx.x == 0 && blockIdx.x == 0 && i < 96) { g_w[i] = w1 - w0; g_w[96 + i] = __builtin_amdgcn_s_memtime() - w1; g_w[192 + i] = rare_cyc; }\n")
src = sub1(src, "__device__ unsigned long long g_tiles[192];\n", "__device__ unsigned long long g_tiles[192];\n__device__ unsigned long long g_w[288];\n")
src = sub1(src, "extern \"C\" int sss_debug_tiles(", "extern \"C\" int sss_debug_w(unsigned long long* host) {\n    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sss::g_w), 288 * 8);\n}\nextern \"C\" int sss_debug_tiles(")
# rare_cyc must be declared before tile_end: hoist
src = sub1(src, "    int n_rare = 0;\n    unsigned long long rare_cyc = 0;\n", "    int n_rare = 0;\n")
src = sub1(src, "    auto tile_end = [&](int i) {", "    unsigned long long rare_cyc = 0;\n    auto tile_end = [&](int i) {")
# tile_top duration of wave 0 (workgroup 0), per tile: g_w[288 + i]; mfma_sub + score_tree durations summed per tile: g_w[384 + i]
src = sub1(src, "    auto tile_top = [&](int i) {                // threshold refresh at the start of tile iteration i > 0\n",
    "    auto tile_top_impl = [&](int i) {\n")
src = sub1(src, "    auto tile_end = [&](int i) {",
    "    auto tile_top = [&](int i) {\n        const unsigned long long tt0 = __builtin_amdgcn_s_memtime();\n        tile_top_impl(i);\n"
    "        if (threadIdx.x == 0 && blockIdx.x == 0 && i < 96) g_w[288 + i] = __builtin_amdgcn_s_memtime() - tt0;\n    };\n    auto tile_end = [&](int i) {")
assert src.count("        mfma_sub(i & 1, sub, next_tile);\n") == 2          # the split loop's and the nested loop's
src = src.replace("        mfma_sub(i & 1, sub, next_tile);\n",
    "        const unsigned long long mm0 = __builtin_amdgcn_s_memtime();\n        mfma_sub(i & 1, sub, next_tile);\n"
    "        if (threadIdx.x == 0 && blockIdx.x == 0 && i < 96) g_w[384 + i] += __builtin_amdgcn_s_memtime() - mm0;\n")
src = sub1(src, "            publish(cls_live);                  // (before the fetch: an atomic behind the two DMA instructions waits for them)\n            tau_fetch();                        // lands under this tile's MFMAs\n",
    "            const unsigned long long pp0 = __builtin_amdgcn_s_memtime();\n            publish(cls_live);\n"
    "            const unsigned long long pp1 = __builtin_amdgcn_s_memtime();\n            tau_fetch();\n"
    "            if (threadIdx.x == 0 && blockIdx.x == 0 && i < 96) { g_w[480 + i] = pp1 - pp0; g_w[576 + i] = __builtin_amdgcn_s_memtime() - pp1; }\n")
src = src.replace("__device__ unsigned long long g_w[288];", "__device__ unsigned long long g_w[672];")
src = src.replace("HIP_SYMBOL(sss::g_w), 288 * 8)", "HIP_SYMBOL(sss::g_w), 672 * 8)")
open(os.path.join(root, "sessionsimilaritysearch_amd/csrc/scan_tl.hip"), "w").write(src)
print("wrote csrc/scan_tl.hip")
