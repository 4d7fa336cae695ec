#include "Camera.h"

namespace acgpt {

void Camera::UVWFrame(float3& U, float3& V, float3& W) const
{
    W = m_lookat - m_eye;                       // focal length lives in |W|
    const float focal = length(W);
    U = normalize(cross(W, m_up));
    V = normalize(cross(U, W));
    const float half_h = focal * tanf(0.5f * m_fovY * kPIf / 180.0f);
    V *= half_h;
    const float half_w = half_h * m_aspectRatio;
    U *= half_w;
}

}  // namespace acgpt
