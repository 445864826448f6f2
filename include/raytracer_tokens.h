/*
 * raytracer_tokens.h -- the drop-in for the reference's device entry point.
 *
 *     extern "C" void launch_kernel(TKSceneData *data, ShadingType stype,
 *                                   int width, int height, uchar4 *output,
 *                                   int sampleCount);            (kernel.h:30-31)
 *
 * Same symbol, same argument order, same contract: the caller owns `output`
 * (width*height*4 bytes, raytrace.cpp:41), the call returns when the frame is
 * in it, a sampleCount without an integer square root prints a message and
 * returns with the frame untouched (kernel.cu:707-712), any device failure
 * prints a message and exits (cudaError.h:6-22).
 *
 * The first argument is a C++ object (std::vector / std::map members), so this
 * is a C++ header.  The TK* types below are layout-compatible with the
 * reference's TokenData.h:11-110 built against GLM 0.9.3 (vec2/vec3 = 2/3
 * packed floats, mat4 = 16 floats column-major) on the same libstdc++: a
 * driver compiled against the reference's own TokenData.h can link this
 * library unchanged.  A driver that includes this header instead gets the same
 * types without GLM.
 */
#ifndef RAYTRACER_TOKENS_H
#define RAYTRACER_TOKENS_H

#include <map>
#include <string>
#include <vector>

#ifndef RT_TOKENS_NO_TYPES   /* define when the reference's TokenData.h / kernel.h are included instead */

struct rt_vec2 { float x, y; rt_vec2() : x(0), y(0) {} rt_vec2(float a, float b) : x(a), y(b) {} };
struct rt_vec3 { float x, y, z; rt_vec3() : x(0), y(0), z(0) {} rt_vec3(float a, float b, float c) : x(a), y(b), z(c) {} explicit rt_vec3(float s) : x(s), y(s), z(s) {} };
struct rt_mat4 {
   float m[16];   /* column-major */
   rt_mat4() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
};

typedef struct TKCamera {
   rt_vec3 pos, up, right, lookAt;
   TKCamera(rt_vec3 position, rt_vec3 upVec, rt_vec3 rightVec, rt_vec3 look)
      : pos(position), up(upVec), right(rightVec), lookAt(look) {}
} TKCamera;

typedef struct TKPigment {
   rt_vec3 clr;
   float f;
   int texId;
   TKPigment(rt_vec3 color = rt_vec3(0.0f), float filter = 0.0f) : clr(color), f(filter), texId(-1) {}
   TKPigment(int nTexId) : clr(rt_vec3(0.0f)), f(0.0f), texId(nTexId) {}
} TKPigment;

typedef struct TKFinish {
   float amb, dif, spec, rough, refl, refr, ior;
   TKFinish() : amb(0.1f), dif(0.6f), spec(0.0), rough(0.05f), refl(0.0f), refr(0.0f), ior(1.0f) {}
   TKFinish(float ambiant, float diffuse, float specular, float roughness, float reflection, float refraction,
            float indexOfRefraction)
      : amb(ambiant), dif(diffuse), spec(specular), rough(roughness), refl(reflection), refr(refraction),
        ior(indexOfRefraction) {}
} TKFinish;

typedef struct TKModifier { TKPigment pig; TKFinish fin; rt_mat4 invTrans; rt_mat4 trans; } TKModifier;
typedef struct TKPointLight { rt_vec3 pos, clr; TKPointLight(rt_vec3 position, rt_vec3 color) : pos(position), clr(color) {} } TKPointLight;
typedef struct TKBox { rt_vec3 p1, p2; TKModifier mod; } TKBox;
typedef struct TKSphere { rt_vec3 p; float r; TKModifier mod; } TKSphere;
typedef struct TKPlane { rt_vec3 n; float d; TKModifier mod; } TKPlane;
typedef struct TKCone { rt_vec3 p1, p2; float r1, r2; TKModifier mod; } TKCone;
typedef struct TKTriangle { rt_vec3 p1, p2, p3; rt_vec2 vt1, vt2, vt3; TKModifier mod; } TKTriangle;
typedef struct TKSmoothTriangle { rt_vec3 p1, p2, p3; rt_vec3 n1, n2, n3; rt_vec2 vt1, vt2, vt3; TKModifier mod; } TKSmoothTriangle;

typedef struct TKSceneData {
   TKCamera *camera;
   std::map<std::string, int> textureMap;
   std::vector<TKPointLight> pointLights;
   std::vector<TKBox> boxes;
   std::vector<TKTriangle> triangles;
   std::vector<TKSmoothTriangle> smoothTriangles;
   std::vector<TKSphere> spheres;
   std::vector<TKPlane> planes;
   TKSceneData() : camera(NULL) {}
} TKSceneData;

typedef enum { PHONG, COOK_TORRANCE } ShadingType;   /* kernel.h:22 */

struct rt_uchar4 { unsigned char x, y, z, w; };      /* CUDA's uchar4 */
#define RT_UCHAR4 rt_uchar4
#else
#define RT_UCHAR4 uchar4
#endif /* RT_TOKENS_NO_TYPES */

/* kernel.h:30-31.  Environment knobs (the reference has none): RT_DEVICE picks
 * the HIP device (default 0); RT_JITTER=0 turns the sub-pixel jitter of
 * generateCameraRays off (default on, like the reference). */
extern "C" void launch_kernel(TKSceneData *data, ShadingType stype, int width, int height, RT_UCHAR4 *output,
                              int sampleCount);

#endif
