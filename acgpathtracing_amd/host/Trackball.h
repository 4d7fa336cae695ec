// Trackball.h — orbit / zoom / fly controller with the reference's sutil::Trackball API
// (sutil/Trackball.h:40-120).  The reference app includes it but never wires it to an
// input callback (PathTracerMain.cpp:18, 686-688); it is kept for API parity.
#pragma once
#include "Camera.h"

namespace acgpt {

class Trackball {
public:
    enum ViewMode { EyeFixed, LookAtFixed };

    bool wheelEvent(int dir);
    void startTracking(int x, int y);
    void updateTracking(int x, int y, int canvasWidth, int canvasHeight);
    void zoom(int direction);
    float moveSpeed() const { return m_moveSpeed; }
    void setMoveSpeed(const float& val) { m_moveSpeed = val; }

    void setCamera(Camera* camera) { m_camera = camera; reinitOrientationFromCamera(); }
    const Camera* currentCamera() const { return m_camera; }
    bool gimbalLock() const { return m_gimbalLock; }
    void setGimbalLock(bool val) { m_gimbalLock = val; }
    void reinitOrientationFromCamera();
    void setReferenceFrame(const float3& u, const float3& v, const float3& w);
    ViewMode viewMode() const { return m_viewMode; }
    void setViewMode(ViewMode val) { m_viewMode = val; }

private:
    void updateCamera();

    bool m_gimbalLock = false;
    ViewMode m_viewMode = LookAtFixed;
    Camera* m_camera = nullptr;
    float m_cameraEyeLookatDistance = 0.0f;
    float m_zoomMultiplier = 1.1f;
    float m_moveSpeed = 1.0f;
    float m_latitude = 0.0f;    // radians
    float m_longitude = 0.0f;   // radians
    int m_prevPosX = 0, m_prevPosY = 0;
    bool m_performTracking = false;
    float3 m_u = {0, 0, 0}, m_v = {0, 0, 0}, m_w = {0, 0, 0};
};

}  // namespace acgpt
