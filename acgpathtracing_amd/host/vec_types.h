// vec_types.h — the few POD vector types and float3 operations the host side needs.
// Operation order follows the reference's sutil/vec_math.h:479-570 (a/s = a*(1/s),
// normalize = v*(1/sqrt(dot)), reflect = i - 2*n*dot(n,i)) so host-computed camera
// frames are bit-identical to the reference's.
#pragma once
#include <cmath>
#include <cstdint>

namespace acgpt {

struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
struct int2 { int x, y; };

inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
inline float3 make_float3(float s) { return make_float3(s, s, s); }
inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }

inline float3 operator-(const float3& a) { return make_float3(-a.x, -a.y, -a.z); }
inline float3 operator+(const float3& a, const float3& b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(const float3& a, const float3& b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator*(const float3& a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
inline float3 operator*(float s, const float3& a) { return make_float3(s * a.x, s * a.y, s * a.z); }
inline void operator*=(float3& a, float s) { a.x *= s; a.y *= s; a.z *= s; }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(const float3& a, const float3& b)
{ return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(const float3& v) { return sqrtf(dot(v, v)); }
inline float3 normalize(const float3& v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }

constexpr float kPIf = 3.14159265358979323846f;     // M_PIf
constexpr float k1_PIf = 0.318309886183790671538f;  // M_1_PIf

}  // namespace acgpt
