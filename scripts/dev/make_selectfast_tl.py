"""Dev helper (not product): csrc/select_tl.hip = csrc/select.hip + phase stamps in k_select_fast (symbol sss_debug_selfast:
[1024 queries][8] s_memtime values of lane 0: start, count known, column sorted, K2 rounds done, re-scored, ranked + written,
end; word 7 = candidate count).  Build:
  cd sessionsimilaritysearch_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c select_tl.hip -o /tmp/select_tl.o &&
  hipcc --offload-arch=gfx950 -shared -o ../../scripts/dev/libsss_sftl.so capi.o ip_topk.o scan.o scan_long.o /tmp/select_tl.o \
        exhaustive.o rowops.o gnn.o vote.o graphbuild.o hamming.o variants.o"""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "sessionsimilaritysearch_amd/csrc/select.hip")).read()
def sub1(s, old, new):
    assert s.count(old) == 1, (s.count(old), old)
    return s.replace(old, new)
a = src.index("__global__ __launch_bounds__(256) void k_select_fast(const SelectArgs A) {")
b = src.index("__global__ __launch_bounds__(SORT_THREADS) void k_select_sort(")
body = src[a:b]
body = sub1(body, "    if (q >= A.nq) return;                                        // whole wave; no block-level sync below\n",
            "    if (q >= A.nq) return;\n    SF(0);\n")
body = sub1(body, "    if (M > FS_CAP) {", "    SF(1);\n    if (lane == 0 && q < 1024) g_sf[q * 8 + 7] = (unsigned long long)M;\n    if (M > FS_CAP) {")
body = sub1(body, "    // ---- K2 rounds: wave-wide arg-max, the owner lane retires its key\n", "    SF(2);\n")
body = sub1(body, "    wave_sync();\n    // ---- float64 re-score: one lane per candidate", "    wave_sync();\n    SF(3);\n    // ---- float64 re-score: one lane per candidate")
body = sub1(body, "    for (int c0 = 0; c0 < K2; c0 += 16) rescore16(sel, resc, c0, K2, A.C, rb, qrow, A.dtype, lane);\n    double qn2 = 0.0;",
            "    for (int c0 = 0; c0 < K2; c0 += 16) rescore16(sel, resc, c0, K2, A.C, rb, qrow, A.dtype, lane);\n    SF(4);\n    double qn2 = 0.0;")
body = sub1(body, "    const int nvalid = *s_nvalid;\n", "    const int nvalid = *s_nvalid;\n    SF(5);\n")
body = sub1(body, "    if (lane == 0) {\n        A.status[q] = st;", "    SF(6);\n    if (lane == 0) {\n        A.status[q] = st;")
body = ("__device__ unsigned long long g_sf[1024 * 8];\n"
        "#define SF(slot) do { if (lane == 0 && q < 1024) g_sf[q * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)\n") + body
src = src[:a] + body + src[b:]
src = sub1(src, "// ------------------------------------------------------------------------------------------\n// k-way merge",
           "extern \"C\" int sss_debug_selfast(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sss::g_sf), 1024 * 8 * 8); }\n\n"
           "// ------------------------------------------------------------------------------------------\n// k-way merge")
open(os.path.join(root, "sessionsimilaritysearch_amd/csrc/select_tl.hip"), "w").write(src)
print("wrote csrc/select_tl.hip")
