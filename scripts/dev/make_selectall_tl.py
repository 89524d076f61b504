"""Dev helper (not product): csrc/select_tl.hip = csrc/select.hip + phase stamps in k_select_all (symbol sss_debug_selall:
[1024 workgroups][8] s_memtime values: start, keys loaded, cut known, survivors compacted, re-scored, sorted/written)."""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "sessionsimilaritysearch_amd/csrc/select.hip")).read()
def sub1(s, old, new):
    assert s.count(old) == 1, (s.count(old), old)
    return s.replace(old, new)
src = sub1(src, "__global__ __launch_bounds__(SORT_THREADS) void k_select_all(const ThrArgs A, int cap_pow2, int second) {\n",
    "__device__ unsigned long long g_sa[1024 * 8];\n#define SA(slot) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && !second) g_sa[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)\n"
    "__global__ __launch_bounds__(SORT_THREADS) void k_select_all(const ThrArgs A, int cap_pow2, int second) {\n    SA(0);\n")
src = sub1(src, "    if (tid == 0) { s_cut = -INFINITY; s_keep = 0u; }\n    __syncthreads();\n", "    if (tid == 0) { s_cut = -INFINITY; s_keep = 0u; }\n    __syncthreads();\n    SA(1);\n")
src = sub1(src, "    const float cut = s_cut;\n", "    SA(2);\n    const float cut = s_cut;\n")
src = sub1(src, "    const int keep = (int)s_keep;                                       // >= k: the k-th largest itself passes the cut\n",
    "    const int keep = (int)s_keep;\n    SA(3);\n    if (threadIdx.x == 0 && blockIdx.x < 1024 && !second) { g_sa[blockIdx.x * 8 + 6] = M; g_sa[blockIdx.x * 8 + 7] = keep; }\n")
src = sub1(src, "    __syncthreads();\n    sort_desc(keys, K2, tid);\n    float* Dq = A.D_out + (size_t)q * k;", "    __syncthreads();\n    SA(4);\n    sort_desc(keys, K2, tid);\n    float* Dq = A.D_out + (size_t)q * k;")
src = sub1(src, "    if (tid == 0) A.status[q] = 0;\n}\n\n// ------------------------------------------------------------------------------------------\n// k-way merge",
    "    if (tid == 0) A.status[q] = 0;\n    SA(5);\n}\nextern \"C\" int sss_debug_selall(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sss::g_sa), 1024 * 8 * 8); }\n\n// ------------------------------------------------------------------------------------------\n// k-way merge")
open(os.path.join(root, "sessionsimilaritysearch_amd/csrc/select_tl.hip"), "w").write(src)
print("wrote csrc/select_tl.hip")
