// Prints sizeof and offsetof of every member of the token structs that cross launch_kernel's boundary.
// Built twice by tests/test_token_layout.py: with -DUSE_REFERENCE_HEADER against the reference's own TokenData.h
// (GLM 0.9.3 from the reference checkout) and without it against include/raytracer_tokens.h; the two listings must
// be equal for a driver compiled against either header to hand the library the same bytes.
#include <cstddef>
#include <cstdio>
#ifdef USE_REFERENCE_HEADER
#include "TokenData.h"
#else
#include "raytracer_tokens.h"
#endif

#define SIZE(T) printf("sizeof(" #T ") = %zu\n", sizeof(T))
#define OFF(T, m) printf("offsetof(" #T ", " #m ") = %zu size %zu\n", offsetof(T, m), sizeof(((T *)0)->m))

int main()
{
   SIZE(TKCamera); OFF(TKCamera, pos); OFF(TKCamera, up); OFF(TKCamera, right); OFF(TKCamera, lookAt);
   SIZE(TKPigment); OFF(TKPigment, clr); OFF(TKPigment, f); OFF(TKPigment, texId);
   SIZE(TKFinish); OFF(TKFinish, amb); OFF(TKFinish, dif); OFF(TKFinish, spec); OFF(TKFinish, rough); OFF(TKFinish, refl);
   OFF(TKFinish, refr); OFF(TKFinish, ior);
   SIZE(TKModifier); OFF(TKModifier, pig); OFF(TKModifier, fin); OFF(TKModifier, invTrans); OFF(TKModifier, trans);
   SIZE(TKPointLight); OFF(TKPointLight, pos); OFF(TKPointLight, clr);
   SIZE(TKBox); OFF(TKBox, p1); OFF(TKBox, p2); OFF(TKBox, mod);
   SIZE(TKSphere); OFF(TKSphere, p); OFF(TKSphere, r); OFF(TKSphere, mod);
   SIZE(TKPlane); OFF(TKPlane, n); OFF(TKPlane, d); OFF(TKPlane, mod);
   SIZE(TKCone); OFF(TKCone, p1); OFF(TKCone, p2); OFF(TKCone, r1); OFF(TKCone, r2); OFF(TKCone, mod);
   SIZE(TKTriangle); OFF(TKTriangle, p1); OFF(TKTriangle, p2); OFF(TKTriangle, p3); OFF(TKTriangle, vt1); OFF(TKTriangle, vt2);
   OFF(TKTriangle, vt3); OFF(TKTriangle, mod);
   SIZE(TKSmoothTriangle); OFF(TKSmoothTriangle, p1); OFF(TKSmoothTriangle, p2); OFF(TKSmoothTriangle, p3); OFF(TKSmoothTriangle, n1);
   OFF(TKSmoothTriangle, n2); OFF(TKSmoothTriangle, n3); OFF(TKSmoothTriangle, vt1); OFF(TKSmoothTriangle, vt2); OFF(TKSmoothTriangle, vt3);
   OFF(TKSmoothTriangle, mod);
   SIZE(TKSceneData); OFF(TKSceneData, camera); OFF(TKSceneData, textureMap); OFF(TKSceneData, pointLights); OFF(TKSceneData, boxes);
   OFF(TKSceneData, triangles); OFF(TKSceneData, smoothTriangles); OFF(TKSceneData, spheres); OFF(TKSceneData, planes);
   // the element order inside the vector and matrix types: x, y, z packed; mat4 column-major, m[col][row]
#ifdef USE_REFERENCE_HEADER
   glm::vec3 v(1.0f, 2.0f, 3.0f); glm::mat4 m(1.0f); m[3][0] = 7.0f; m[0][3] = 9.0f;
   const float *pv = (const float *)&v, *pm = (const float *)&m;
#else
   rt_vec3 v(1.0f, 2.0f, 3.0f); rt_mat4 m; m.m[12] = 7.0f; m.m[3] = 9.0f;
   const float *pv = (const float *)&v, *pm = (const float *)&m;
#endif
   printf("vec3 as floats: %g %g %g\n", pv[0], pv[1], pv[2]);
   printf("mat4 as floats:");
   for (int i = 0; i < 16; i++) printf(" %g", pm[i]);
   printf("\n");
   return 0;
}
