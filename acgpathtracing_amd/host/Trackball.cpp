// Trackball.cpp — behaviour of sutil/Trackball.cpp:51-137 (latitude/longitude orbit with
// 0.5 degree per pixel, latitude clamped to +-89 degrees, zoom factor 1.1).
#include "Trackball.h"
#include <algorithm>
#include <cmath>
#include <utility>

namespace acgpt {

static inline float to_radians(float deg) { return deg * kPIf / 180.0f; }
static inline float to_degrees(float rad) { return rad * k1_PIf * 180.0f; }

void Trackball::startTracking(int x, int y)
{
    m_prevPosX = x;
    m_prevPosY = y;
    m_performTracking = true;
}

void Trackball::updateTracking(int x, int y, int, int)
{
    if (!m_performTracking) { startTracking(x, y); return; }
    const int dx = x - m_prevPosX, dy = y - m_prevPosY;
    m_prevPosX = x;
    m_prevPosY = y;
    m_latitude = to_radians(std::min(89.0f, std::max(-89.0f, to_degrees(m_latitude) + 0.5f * dy)));
    m_longitude = to_radians(fmod(to_degrees(m_longitude) - 0.5f * dx, 360.0f));
    updateCamera();
    if (!m_gimbalLock) {
        reinitOrientationFromCamera();
        m_camera->setUp(m_w);
    }
}

void Trackball::updateCamera()
{
    float3 local;
    local.x = cos(m_latitude) * sin(m_longitude);
    local.y = cos(m_latitude) * cos(m_longitude);
    local.z = sin(m_latitude);
    const float3 dirWS = m_u * local.x + m_v * local.y + m_w * local.z;
    if (m_viewMode == EyeFixed) {
        const float3& eye = m_camera->eye();
        m_camera->setLookat(eye - dirWS * m_cameraEyeLookatDistance);
    } else {
        const float3& lookat = m_camera->lookat();
        m_camera->setEye(lookat + dirWS * m_cameraEyeLookatDistance);
    }
}

void Trackball::setReferenceFrame(const float3& u, const float3& v, const float3& w)
{
    m_u = u; m_v = v; m_w = w;
    const float3 dirWS = -normalize(m_camera->lookat() - m_camera->eye());
    const float lx = dot(dirWS, u), ly = dot(dirWS, v), lz = dot(dirWS, w);
    m_longitude = atan2(lx, ly);
    m_latitude = asin(lz);
}

void Trackball::zoom(int direction)
{
    const float z = (direction > 0) ? 1 / m_zoomMultiplier : m_zoomMultiplier;
    m_cameraEyeLookatDistance *= z;
    const float3& lookat = m_camera->lookat();
    const float3& eye = m_camera->eye();
    m_camera->setEye(lookat + (eye - lookat) * z);
}

void Trackball::reinitOrientationFromCamera()
{
    m_camera->UVWFrame(m_u, m_v, m_w);
    m_u = normalize(m_u);
    m_v = normalize(m_v);
    m_w = normalize(-m_w);
    std::swap(m_v, m_w);
    m_latitude = 0.0f;
    m_longitude = 0.0f;
    m_cameraEyeLookatDistance = length(m_camera->lookat() - m_camera->eye());
}

bool Trackball::wheelEvent(int dir)
{
    zoom(dir);
    return true;
}

}  // namespace acgpt
