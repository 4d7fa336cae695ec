// Camera.h — pinhole camera with the reference's sutil::Camera API
// (sutil/Camera.h:38-75): same constructor defaults, accessors and UVWFrame().
#pragma once
#include "vec_types.h"

namespace acgpt {

class Camera {
public:
    Camera() : m_eye(make_float3(1.0f)), m_lookat(make_float3(0.0f)), m_up(make_float3(0.0f, 1.0f, 0.0f)), m_fovY(35.0f), m_aspectRatio(1.0f) {}
    Camera(const float3& eye, const float3& lookat, const float3& up, float fovY, float aspectRatio)
        : m_eye(eye), m_lookat(lookat), m_up(up), m_fovY(fovY), m_aspectRatio(aspectRatio) {}

    float3 direction() const { return normalize(m_lookat - m_eye); }
    void setDirection(const float3& dir) { m_lookat = m_eye + length(m_lookat - m_eye) * dir; }

    const float3& eye() const { return m_eye; }
    void setEye(const float3& val) { m_eye = val; }
    const float3& lookat() const { return m_lookat; }
    void setLookat(const float3& val) { m_lookat = val; }
    const float3& up() const { return m_up; }
    void setUp(const float3& val) { m_up = val; }
    const float& fovY() const { return m_fovY; }
    void setFovY(const float& val) { m_fovY = val; }
    const float& aspectRatio() const { return m_aspectRatio; }
    void setAspectRatio(const float& val) { m_aspectRatio = val; }

    // U, V, W are orthogonal but NOT normalised: |W| is the focal distance,
    // |V| = |W| tan(fovY/2), |U| = |V| * aspect (sutil/Camera.cpp:34-45).
    void UVWFrame(float3& U, float3& V, float3& W) const;

private:
    float3 m_eye, m_lookat, m_up;
    float m_fovY, m_aspectRatio;
};

}  // namespace acgpt
