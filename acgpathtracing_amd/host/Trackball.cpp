// Trackball.cpp — see Trackball.h for the behaviour this implements.  The arithmetic (degree <-> radian
// round trips in fp32, the libm calls and their argument types) is what
// tests/test_host_golden.py::test_trackball_matches_reference holds bit for bit against the reference's
// sutil/Trackball.cpp:51-137 compiled in the authoring container.
#include "Trackball.h"
#include <algorithm>
#include <cmath>

namespace acgpt {

namespace {
constexpr float kDegPerPixel = 0.5f;
constexpr float kLatLimitDeg = 89.0f;

inline float as_degrees(float rad) { return rad * k1_PIf * 180.0f; }
inline float as_radians(float deg) { return deg * kPIf / 180.0f; }
inline float clamp_lat(float deg) { return std::min(kLatLimitDeg, std::max(-kLatLimitDeg, deg)); }
}  // namespace

void Trackball::startTracking(int x, int y)
{
    pointer_.x = x;
    pointer_.y = y;
    pointer_.dragging = true;
}

void Trackball::updateTracking(int x, int y, int /*canvasWidth*/, int /*canvasHeight*/)
{
    if (!pointer_.dragging) {       // first event of a drag only anchors the pointer
        startTracking(x, y);
        return;
    }
    const int moved_x = x - pointer_.x;
    const int moved_y = y - pointer_.y;
    pointer_.x = x;
    pointer_.y = y;

    pos_.lat = as_radians(clamp_lat(as_degrees(pos_.lat) + kDegPerPixel * moved_y));
    pos_.lon = as_radians(fmod(as_degrees(pos_.lon) - kDegPerPixel * moved_x, 360.0f));
    placeCamera();

    if (!frame_locked_) {           // free exploration: the orbit frame follows the camera
        reinitOrientationFromCamera();
        cam_->setUp(axes_[kPole]);
    }
}

float3 Trackball::offsetOnSphere() const
{
    const float on_right   = cos(pos_.lat) * sin(pos_.lon);
    const float on_forward = cos(pos_.lat) * cos(pos_.lon);
    const float on_pole    = sin(pos_.lat);
    return axes_[kRight] * on_right + axes_[kForward] * on_forward + axes_[kPole] * on_pole;
}

void Trackball::placeCamera()
{
    const float3 offset = offsetOnSphere();
    if (mode_ == LookAtFixed) cam_->setEye(cam_->lookat() + offset * radius_);
    else                      cam_->setLookat(cam_->eye() - offset * radius_);
}

void Trackball::setReferenceFrame(const float3& u, const float3& v, const float3& w)
{
    axes_[kRight] = u;
    axes_[kForward] = v;
    axes_[kPole] = w;
    // where the eye sits on the sphere of the new frame
    const float3 towards_eye = -normalize(cam_->lookat() - cam_->eye());
    const float cr = dot(towards_eye, u), cf = dot(towards_eye, v), cp = dot(towards_eye, w);
    pos_.lon = atan2(cr, cf);
    pos_.lat = asin(cp);
}

void Trackball::zoom(int direction)
{
    const float scale = direction > 0 ? 1 / wheel_factor_ : wheel_factor_;
    radius_ *= scale;
    const float3 pivot = cam_->lookat();
    cam_->setEye(pivot + (cam_->eye() - pivot) * scale);
}

bool Trackball::wheelEvent(int dir)
{
    zoom(dir);
    return true;
}

void Trackball::reinitOrientationFromCamera()
{
    // camera frame U (right), V (up), W (view direction) -> orbit frame {right, towards the eye, up}:
    // the camera's up becomes the pole and the sphere position restarts at the origin of the new frame
    float3 cam_u, cam_v, cam_w;
    cam_->UVWFrame(cam_u, cam_v, cam_w);
    axes_[kRight] = normalize(cam_u);
    axes_[kForward] = normalize(-cam_w);
    axes_[kPole] = normalize(cam_v);
    pos_ = SpherePos();
    radius_ = length(cam_->lookat() - cam_->eye());
}

}  // namespace acgpt
