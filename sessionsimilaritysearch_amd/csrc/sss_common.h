// Shared device/host helpers for libsss (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SSS_OK 0
#define SSS_EINVAL (-1)
#define SSS_EWORKSPACE (-2)
#define SSS_EHIP (-3)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace sss {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// per-device one-time setup flags (hipFuncSetAttribute is per device); defined in scan.hip
constexpr int MAX_DEVICES = 64;
int current_device();

// float -> uint32 whose unsigned order equals the float order (-inf lowest, +inf highest).
__device__ __host__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ __forceinline__ float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, u);
}
// 64-bit selection key: larger key == better candidate == (higher score, then lower id).
// id -1 (empty slot) with score -inf gives the smallest key any slot can have; 0 is below all.
__device__ __host__ __forceinline__ uint64_t make_key(float s, int32_t id) {
    return ((uint64_t)f2ord(s) << 32) | (uint32_t)(~(uint32_t)id);
}
__device__ __host__ __forceinline__ float key_score(uint64_t k) { return ord2f((uint32_t)(k >> 32)); }
__device__ __host__ __forceinline__ int32_t key_id(uint64_t k) { return (int32_t)(~(uint32_t)k); }

}  // namespace sss
