/*
 * camera.cpp -- orbit camera about the world origin (the reference's
 * ModelCamera, src/renderer/camera.cpp:7-141), headless: the GLFW button and
 * cursor polls of glfwHandleCursor become arguments of handleCursor().
 */
#include <cmath>

#include "bdpt_host.h"

namespace bdpt_host {

static vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
static vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

/* rotation of v about `axis` by `degrees` (Rodrigues); the reference calls
 * glm::rotate with what it names "degree space" angles, camera.cpp:111-121 */
static vec3 rotate(vec3 v, float degrees, vec3 axis)
{
    const float rad = degrees * 0.01745329251994329577f;
    const float c = std::cos(rad), s = std::sin(rad);
    const vec3 a = normalize(axis);
    return v * c + cross(a, v) * s + a * (dot(a, v) * (1.0f - c));
}

ModelCamera::ModelCamera() {}

void ModelCamera::reset(vec3 vDir)
{
    vPos = normalize(vDir) * fRadius;                                    /* camera.cpp:45 */
    vLookAt = normalize(-vPos);                                          /* :47 */
    vUp = normalize(cross(vLookAt, cross(vLookAt, vOrientation)));       /* :50 */
}

void ModelCamera::handleCursor(bool left_down, bool middle_down, double dx, double dy)
{
    if (scrollState == SCROLL_INACTIVE) {                                /* camera.cpp:70-75 */
        if (middle_down) scrollState = SCROLL_ACTIVE;
    } else {                                                             /* :77-89 */
        if (!middle_down) {
            scrollState = SCROLL_INACTIVE;
        } else {
            fRadius -= (float)(dy * fScrollSpeed);
            reset(getEye());
        }
    }
    if (iState == CAM_INACTIVE) {                                        /* :91-97 */
        if (left_down) iState = CAM_ACTIVE;
        return;
    }
    if (!left_down) {                                                    /* :98-104 */
        iState = CAM_INACTIVE;
        return;
    }
    const double degX = fSpeedX * dx, degY = fSpeedY * dy;               /* :112-113 */
    vec3 newPos = vPos;
    newPos = rotate(newPos, (float)degX, vUp);                           /* :120 */
    newPos = rotate(newPos, (float)degY, cross(vUp, vLookAt));           /* :121 */
    vPos = newPos;
    vLookAt = normalize(-vPos);                                          /* :127 */
    vUp = normalize(cross(vLookAt, cross(vLookAt, vOrientation)));       /* :130 */
}

void ModelCamera::setPose(const bdpt_camera &c)
{
    vPos = {c.pos[0], c.pos[1], c.pos[2]};
    vLookAt = {c.look_at[0], c.look_at[1], c.look_at[2]};
    vUp = {c.up[0], c.up[1], c.up[2]};
    fRadius = std::sqrt(vPos.x * vPos.x + vPos.y * vPos.y + vPos.z * vPos.z);
}

bdpt_camera ModelCamera::getCLCamera() const
{
    bdpt_camera c;
    c.pos[0] = vPos.x; c.pos[1] = vPos.y; c.pos[2] = vPos.z; c.pos[3] = 0.0f;
    c.look_at[0] = vLookAt.x; c.look_at[1] = vLookAt.y; c.look_at[2] = vLookAt.z; c.look_at[3] = 0.0f;
    c.up[0] = vUp.x; c.up[1] = vUp.y; c.up[2] = vUp.z; c.up[3] = 0.0f;
    return c;
}

}  // namespace bdpt_host
