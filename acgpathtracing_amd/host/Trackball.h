// Trackball.h — orbit / zoom camera controller offering the public API of the reference's
// sutil::Trackball (sutil/Trackball.h:40-95), written from its behaviour:
//   * the controller holds an orbit frame (three world-space axes; the third is the pole), a position on
//     the unit sphere of that frame as latitude / longitude, and the eye-lookat distance;
//   * a pointer drag of (dx, dy) pixels moves the position by (-dx/2, +dy/2) degrees, latitude clamped to
//     [-89, 89], longitude wrapped with fmod 360;
//   * the camera's eye (LookAtFixed) or lookat (EyeFixed) is then placed on that sphere;
//   * without gimbal lock the orbit frame is re-derived from the camera after every drag;
//   * a wheel step scales the distance by 1.1 or 1/1.1.
// The reference app includes the class but never wires it to an input callback (PathTracerMain.cpp:18,
// 686-688); acgpt_main drives it from --orbit / --zoom.
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
    float moveSpeed() const { return fly_speed_; }
    void setMoveSpeed(const float& val) { fly_speed_ = val; }

    void setCamera(Camera* camera) { cam_ = camera; reinitOrientationFromCamera(); }
    const Camera* currentCamera() const { return cam_; }
    bool gimbalLock() const { return frame_locked_; }
    void setGimbalLock(bool val) { frame_locked_ = val; }
    void reinitOrientationFromCamera();
    void setReferenceFrame(const float3& u, const float3& v, const float3& w);
    ViewMode viewMode() const { return mode_; }
    void setViewMode(ViewMode val) { mode_ = val; }

private:
    struct Pointer { int x = 0, y = 0; bool dragging = false; };
    struct SpherePos { float lat = 0.0f, lon = 0.0f; };      // radians
    enum Axis { kRight = 0, kForward = 1, kPole = 2 };

    float3 offsetOnSphere() const;      // unit vector from the fixed point towards the moving one, world space
    void placeCamera();                 // put eye / lookat where the sphere position says

    Camera*   cam_ = nullptr;
    ViewMode  mode_ = LookAtFixed;
    bool      frame_locked_ = false;
    Pointer   pointer_;
    SpherePos pos_;
    float3    axes_[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    float     radius_ = 0.0f;           // |lookat - eye|
    float     wheel_factor_ = 1.1f;
    float     fly_speed_ = 1.0f;
};

}  // namespace acgpt
