/*
 * rt_pov.cpp -- POV-Ray-subset scene parser: the host step in front of the
 * render path.  Accepts what POVRayParser::parseFile (POVRayParser.cpp:14-62)
 * accepts and produces the same numbers bit for bit:
 *   - objects: camera light_source area_light box sphere cone plane triangle
 *     smooth_triangle (POVRayParser.cpp:33-55); a cone is parsed and dropped
 *     (POVRayParser.cpp:229-268);
 *   - modifiers scale / rotate / translate are LEFT-multiplied onto the
 *     object's matrix in file order (POVRayParser.cpp:584, 604-606, 621),
 *     rotate is in degrees about X, then Y, then Z;
 *   - inv_trans is GLM's cofactor inverse (glm/core/func_matrix.inl:523-583),
 *     restated here with the same operation order so that the matrices, and
 *     with them every intersection, come out identical;
 *   - tokens are whitespace-separated words as `istream >> string` sees them
 *     (\r counts as whitespace, so CR-only files parse), vectors are read up
 *     to '>' and decoded with sscanf, exactly like the reference.
 * No iostreams: the file is scanned in memory.
 */
#include "rt_common.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

const size_t kBufferSize = 200;   /* POVRayParser.h:17 */

/* ---- 4x4 matrices, column-major c[col][row], GLM operation order -------- */
struct V4 { float x, y, z, w; };
struct M4 { V4 c[4]; };

inline V4 v4(float x, float y, float z, float w) { V4 r = {x, y, z, w}; return r; }
inline V4 operator*(V4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline V4 operator*(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline V4 operator+(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline V4 operator-(V4 a, V4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline V4 operator/(V4 a, float s) { return v4(a.x / s, a.y / s, a.z / s, a.w / s); }
inline float elem(const V4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; }

M4 identity()
{
   M4 m;
   m.c[0] = v4(1, 0, 0, 0); m.c[1] = v4(0, 1, 0, 0); m.c[2] = v4(0, 0, 1, 0); m.c[3] = v4(0, 0, 0, 1);
   return m;
}
/* type_mat4x4.inl:757-779 */
M4 mul(const M4 &a, const M4 &b)
{
   M4 r;
   for (int k = 0; k < 4; k++)
      r.c[k] = a.c[0] * b.c[k].x + a.c[1] * b.c[k].y + a.c[2] * b.c[k].z + a.c[3] * b.c[k].w;
   return r;
}
/* gtc/matrix_transform.inl:80-93 on the identity */
M4 scale_matrix(float x, float y, float z)
{
   M4 m = identity(), r;
   r.c[0] = m.c[0] * x; r.c[1] = m.c[1] * y; r.c[2] = m.c[2] * z; r.c[3] = m.c[3];
   return r;
}
/* gtc/matrix_transform.inl:31-41 on the identity */
M4 translate_matrix(float x, float y, float z)
{
   M4 m = identity(), r = m;
   r.c[3] = m.c[0] * x + m.c[1] * y + m.c[2] * z + m.c[3];
   return r;
}
/* gtc/matrix_transform.inl:43-78 on the identity; angle in degrees (func_trigonometric.inl:35-44) */
M4 rotate_matrix(float angle, float vx, float vy, float vz)
{
   const float pi = float(3.1415926535897932384626433832795);
   float a = angle * (pi / float(180));
   float c = std::cos(a);
   float s = std::sin(a);
   float sqr = vx * vx + vy * vy + vz * vz;
   float inv = float(1) / std::sqrt(sqr);
   float axis[3] = {vx * inv, vy * inv, vz * inv};
   float temp[3] = {(float(1) - c) * axis[0], (float(1) - c) * axis[1], (float(1) - c) * axis[2]};
   float R[3][3];
   R[0][0] = c + temp[0] * axis[0];
   R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
   R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
   R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
   R[1][1] = c + temp[1] * axis[1];
   R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
   R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
   R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
   R[2][2] = c + temp[2] * axis[2];
   M4 m = identity(), r;
   r.c[0] = m.c[0] * R[0][0] + m.c[1] * R[0][1] + m.c[2] * R[0][2];
   r.c[1] = m.c[0] * R[1][0] + m.c[1] * R[1][1] + m.c[2] * R[1][2];
   r.c[2] = m.c[0] * R[2][0] + m.c[1] * R[2][1] + m.c[2] * R[2][2];
   r.c[3] = m.c[3];
   return r;
}
/* func_matrix.inl:523-583 */
M4 inverse(const M4 &mm)
{
#define M(c_, r_) elem(mm.c[c_], r_)
   float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
   float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
   float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
   float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
   float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
   float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
   float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
   float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
   float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
   float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
   float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
   float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
   float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
   float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
   float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
   float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
   float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
   float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

   V4 SignA = v4(+1, -1, +1, -1), SignB = v4(-1, +1, -1, +1);
   V4 Fac0 = v4(Coef00, Coef00, Coef02, Coef03);
   V4 Fac1 = v4(Coef04, Coef04, Coef06, Coef07);
   V4 Fac2 = v4(Coef08, Coef08, Coef10, Coef11);
   V4 Fac3 = v4(Coef12, Coef12, Coef14, Coef15);
   V4 Fac4 = v4(Coef16, Coef16, Coef18, Coef19);
   V4 Fac5 = v4(Coef20, Coef20, Coef22, Coef23);
   V4 Vec0 = v4(M(1, 0), M(0, 0), M(0, 0), M(0, 0));
   V4 Vec1 = v4(M(1, 1), M(0, 1), M(0, 1), M(0, 1));
   V4 Vec2 = v4(M(1, 2), M(0, 2), M(0, 2), M(0, 2));
   V4 Vec3 = v4(M(1, 3), M(0, 3), M(0, 3), M(0, 3));
#undef M
   V4 Inv0 = SignA * (Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2);
   V4 Inv1 = SignB * (Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4);
   V4 Inv2 = SignA * (Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5);
   V4 Inv3 = SignB * (Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5);
   V4 Row0 = v4(Inv0.x, Inv1.x, Inv2.x, Inv3.x);
   const V4 &m0 = mm.c[0];
   float det = m0.x * Row0.x + m0.y * Row0.y + m0.z * Row0.z + m0.w * Row0.w;
   M4 r;
   r.c[0] = Inv0 / det; r.c[1] = Inv1 / det; r.c[2] = Inv2 / det; r.c[3] = Inv3 / det;
   return r;
}
void store(float *dst, const M4 &m)
{
   for (int c = 0; c < 4; c++) { dst[c * 4 + 0] = m.c[c].x; dst[c * 4 + 1] = m.c[c].y; dst[c * 4 + 2] = m.c[c].z; dst[c * 4 + 3] = m.c[c].w; }
}

/* ---- scanner with the extraction rules of std::istream ------------------ */
struct Scanner {
   std::string data;
   size_t pos = 0;
   bool failed = false;

   static bool space(char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); }
   bool at_end() const { return pos >= data.size(); }
   void skip_ws() { while (pos < data.size() && space(data[pos])) pos++; }

   /* in >> std::string */
   std::string word()
   {
      skip_ws();
      size_t b = pos;
      while (pos < data.size() && !space(data[pos])) pos++;
      return data.substr(b, pos - b);
   }
   /* in >> word, then everything but its first character is put back (parseCharacter, POVRayParser.cpp:671-689) */
   bool character(char want)
   {
      skip_ws();
      if (at_end() || data[pos] != want) return false;
      pos++;
      return true;
   }
   /* looks at the next word without consuming it (isComment / isUV, POVRayParser.cpp:692-735);
    * the leading whitespace stays consumed, as it does there */
   std::string peek_word()
   {
      skip_ws();
      size_t save = pos;
      std::string w = word();
      pos = save;
      return w;
   }
   /* in.getline(buffer, kBufferSize, delim): up to kBufferSize-1 characters, the delimiter is consumed */
   bool getline(std::string &out, char delim)
   {
      out.clear();
      while (pos < data.size()) {
         char ch = data[pos];
         if (ch == delim) { pos++; return true; }
         if (out.size() + 1 >= kBufferSize) { failed = true; return false; }   /* the reference's stream would go bad here */
         out.push_back(ch); pos++;
      }
      return !out.empty();   /* hit end of file: the stream still hands back what it read */
   }
   /* in >> float: the longest prefix that reads as a decimal number */
   bool number(float &v)
   {
      skip_ws();
      size_t p = pos;
      auto digit = [&](size_t i) { return i < data.size() && data[i] >= '0' && data[i] <= '9'; };
      if (p < data.size() && (data[p] == '+' || data[p] == '-')) p++;
      size_t digits = 0;
      while (digit(p)) { p++; digits++; }
      if (p < data.size() && data[p] == '.') { p++; while (digit(p)) { p++; digits++; } }
      if (!digits) { failed = true; return false; }
      if (p < data.size() && (data[p] == 'e' || data[p] == 'E')) {
         size_t q = p + 1;
         if (q < data.size() && (data[q] == '+' || data[q] == '-')) q++;
         if (digit(q)) { while (digit(q)) q++; p = q; }
      }
      std::string tok = data.substr(pos, p - pos);
      v = strtof(tok.c_str(), NULL);
      pos = p;
      return true;
   }
   void skip_comments()
   {
      /* checkComments, POVRayParser.cpp:737-741 */
      for (;;) {
         std::string w = peek_word();
         if (w.size() >= 2 && w[0] == '/' && w[1] == '/') {
            while (pos < data.size() && data[pos] != '\n') pos++;
            if (pos < data.size()) pos++;
         } else break;
      }
   }
};

struct SceneOut {
   rt_camera cam;
   bool has_camera = false;
   std::vector<rt_sphere> spheres;
   std::vector<rt_triangle> triangles;
   std::vector<rt_box> boxes;
   std::vector<rt_smooth_triangle> smooth;
   std::vector<rt_plane> planes;
   std::vector<rt_light> lights;
   std::map<std::string, int> textures;
};

void default_modifier(rt_modifier *m)
{
   memset(m, 0, sizeof *m);
   /* TKPigment() and TKFinish() defaults, TokenData.h:24-25, 36-38 */
   m->mat.amb = 0.1f; m->mat.dif = 0.6f; m->mat.spec = 0.0f; m->mat.rough = 0.05f;
   m->mat.refl = 0.0f; m->mat.refr = 0.0f; m->mat.ior = 1.0f; m->mat.tex_id = -1;
   store(m->trans, identity()); store(m->inv_trans, identity());
}

bool fail(const char *what) { rt_set_error("parse error: %s", what); return false; }

/* parseVector, POVRayParser.cpp:645-658 */
bool parse_vector(Scanner &in, float *v)
{
   std::string buf;
   in.getline(buf, '>');
   if (sscanf(buf.c_str(), " < %f , %f , %f > ", &v[0], &v[1], &v[2]) == 3) return true;
   return fail("Bad format for vector entry");
}
/* parseUV, POVRayParser.cpp:630-643 */
bool parse_uv(Scanner &in, float *uv)
{
   std::string buf; float a, b;
   in.getline(buf, '>');
   if (sscanf(buf.c_str(), " < %f , %f > ", &a, &b) == 2) { uv[0] = a; uv[1] = b; return true; }
   return fail("Bad format for uv entry");
}
bool parse_word(Scanner &in, const char *s)
{
   std::string w = in.word();
   if (w != s) { rt_set_error("parse error: expected \"%s\", found \"%s\"", s, w.c_str()); return false; }
   return true;
}
bool parse_char(Scanner &in, char c)
{
   if (!in.character(c)) { rt_set_error("parse error: expected \"%c\"", c); return false; }
   return true;
}

/* parseFinish, POVRayParser.cpp:512-576 */
bool parse_finish(Scanner &in, rt_material *m)
{
   std::string w = in.word();
   if (w.empty() || w[0] != '{') return fail("expected \"{\" while parsing finish");
   w = w.substr(1);
   for (;;) {
      if (!w.empty() && w[0] == '}') break;
      float *dst = NULL;
      if (w.empty()) { /* "{ ambient": nothing glued to the brace */ }
      else if (w == "ambient") dst = &m->amb;
      else if (w == "diffuse") dst = &m->dif;
      else if (w == "specular") dst = &m->spec;
      else if (w == "roughness") dst = &m->rough;
      else if (w == "reflection") dst = &m->refl;
      else if (w == "refraction") dst = &m->refr;
      else if (w == "ior") dst = &m->ior;
      else return fail("Bad formatting in finish");
      if (dst && !in.number(*dst)) return fail("number expected in finish");
      if (in.at_end()) return fail("unterminated finish");
      w = in.word();
      if (w.empty()) return fail("unterminated finish");
   }
   /* whatever was glued behind the '}' goes back into the stream */
   in.pos -= (w.size() - 1);
   return true;
}

/* parsePigment, POVRayParser.cpp:465-509 */
bool parse_pigment(Scanner &in, rt_material *m, SceneOut &out)
{
   std::string buf;
   in.getline(buf, '}');
   float c[3] = {0, 0, 0}, f = 0;
   if (sscanf(buf.c_str(), " { color rgb < %f , %f , %f > ", &c[0], &c[1], &c[2]) == 3) {
      m->clr[0] = c[0]; m->clr[1] = c[1]; m->clr[2] = c[2]; m->alpha = 0.0f; m->tex_id = -1;
      return true;
   }
   if (sscanf(buf.c_str(), " { color rgbf < %f , %f , %f , %f > ", &c[0], &c[1], &c[2], &f) == 4) {
      m->clr[0] = c[0]; m->clr[1] = c[1]; m->clr[2] = c[2]; m->alpha = f; m->tex_id = -1;
      return true;
   }
   char name[kBufferSize + 1];
   if (sscanf(buf.c_str(), " { image_map \"%200s\" ", name) == 1) {
      int id;
      std::map<std::string, int>::iterator it = out.textures.find(name);
      if (it == out.textures.end()) { id = (int)out.textures.size(); out.textures[name] = id; }
      else id = it->second;
      m->clr[0] = m->clr[1] = m->clr[2] = 0.0f; m->alpha = 0.0f; m->tex_id = id;
      return true;
   }
   return fail("Bad format found for pigment");
}

/* parseModifiers, POVRayParser.cpp:432-463 */
bool parse_modifiers(Scanner &in, rt_modifier *mod, SceneOut &out)
{
   M4 stack = identity();
   std::string w = in.word();
   while (w != "}") {
      float v[3];
      if (w == "scale") { if (!parse_vector(in, v)) return false; stack = mul(scale_matrix(v[0], v[1], v[2]), stack); }
      else if (w == "rotate") {
         if (!parse_vector(in, v)) return false;
         stack = mul(rotate_matrix(v[0], 1.0f, 0.0f, 0.0f), stack);
         stack = mul(rotate_matrix(v[1], 0.0f, 1.0f, 0.0f), stack);
         stack = mul(rotate_matrix(v[2], 0.0f, 0.0f, 1.0f), stack);
      }
      else if (w == "translate") { if (!parse_vector(in, v)) return false; stack = mul(translate_matrix(v[0], v[1], v[2]), stack); }
      else if (w == "finish") { if (!parse_finish(in, &mod->mat)) return false; }
      else if (w == "pigment") { if (!parse_pigment(in, &mod->mat, out)) return false; }
      else return fail("Invalid format for modifiers");
      if (in.failed) return fail("line too long");
      if (in.at_end()) return fail("unterminated object");
      w = in.word();
   }
   store(mod->trans, stack);
   store(mod->inv_trans, inverse(stack));
   return true;
}

bool maybe_uv(Scanner &in, float *vt1, float *vt2, float *vt3)
{
   if (in.at_end()) return true;
   std::string w = in.peek_word();
   if (w.size() >= 2 && w[0] == 'u' && w[1] == 'v') {
      return parse_word(in, "uv") && parse_char(in, '{') && parse_uv(in, vt1) && parse_char(in, ',') && parse_uv(in, vt2) &&
             parse_char(in, ',') && parse_uv(in, vt3) && parse_char(in, '}');
   }
   return true;
}

bool parse_file(Scanner &in, SceneOut &out)
{
   while (!in.at_end()) {
      in.skip_comments();
      std::string first = in.word();
      if (first.empty()) continue;
      if (first == "light_source") {
         /* POVRayParser.cpp:64-84 */
         std::string buf; float p[3], c[3];
         in.getline(buf, '}');
         if (sscanf(buf.c_str(), " { < %f , %f , %f > color rgb < %f , %f , %f > ", &p[0], &p[1], &p[2], &c[0], &c[1], &c[2]) != 6)
            return fail("Invalid format for light_source");
         rt_light l; memcpy(l.pos, p, 12); memcpy(l.clr, c, 12);
         out.lights.push_back(l);
      } else if (first == "area_light") {
         /* POVRayParser.cpp:86-121: samples x samples point lights, colour divided by samples^2 */
         std::string buf; float v1[3], v2[3], v3[3], c[3]; int samples;
         in.getline(buf, '}');
         if (sscanf(buf.c_str(), " { < %f , %f , %f > < %f , %f , %f > < %f , %f , %f > sample %d color rgb < %f , %f , %f > ",
                    &v1[0], &v1[1], &v1[2], &v2[0], &v2[1], &v2[2], &v3[0], &v3[1], &v3[2], &samples, &c[0], &c[1], &c[2]) != 13)
            return fail("Invalid format for area_light");
         /* the reference loops whatever the count says (and divides the colour by zero for 0); a count that would make
          * more than 2^24 lights is refused here instead of being left to run out of memory */
         if (samples > 4096) return fail("area_light: sample count too large");
         c[0] /= samples * samples; c[1] /= samples * samples; c[2] /= samples * samples;
         float e1[3] = {v3[0] - v2[0], v3[1] - v2[1], v3[2] - v2[2]};
         float e2[3] = {v1[0] - v2[0], v1[1] - v2[1], v1[2] - v2[2]};
         for (int x = 0; x < samples; x++)
            for (int y = 0; y < samples; y++) {
               float fx = x / (float)samples, fy = y / (float)samples;
               rt_light l;
               for (int k = 0; k < 3; k++) { l.pos[k] = v2[k] + e1[k] * fx + e2[k] * fy; l.clr[k] = c[k]; }
               out.lights.push_back(l);
            }
      } else if (first == "camera") {
         /* POVRayParser.cpp:123-167 */
         if (!parse_char(in, '{')) return false;
         rt_camera cam; memset(&cam, 0, sizeof cam);
         std::string w = in.word();
         while (w != "}") {
            std::string buf;
            in.getline(buf, '>');
            float *dst = w == "location" ? cam.pos : w == "up" ? cam.up : w == "right" ? cam.right : w == "look_at" ? cam.look_at : NULL;
            float v[3];
            if (!dst || sscanf(buf.c_str(), " < %f , %f , %f ", &v[0], &v[1], &v[2]) != 3) return fail("Invalid format for camera");
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
            in.skip_comments();
            if (in.at_end()) return fail("unterminated camera");
            w = in.word();
         }
         out.cam = cam; out.has_camera = true;
      } else if (first == "box") {
         /* POVRayParser.cpp:169-199 */
         rt_box b; default_modifier(&b.mod);
         if (!parse_char(in, '{') || !parse_vector(in, b.p1) || !parse_word(in, ",") || !parse_vector(in, b.p2)) return false;
         bool ok = parse_modifiers(in, &b.mod, out);
         out.boxes.push_back(b);
         if (!ok) return false;
      } else if (first == "sphere") {
         /* POVRayParser.cpp:201-227; the brace and the comma are consumed but, as there, never checked */
         rt_sphere s; default_modifier(&s.mod);
         (void)in.character('{');
         if (!parse_vector(in, s.c)) return false;
         (void)in.character(',');
         if (!in.number(s.r)) return fail("sphere radius expected");
         bool ok = parse_modifiers(in, &s.mod, out);
         out.spheres.push_back(s);
         if (!ok) return false;
      } else if (first == "cone") {
         /* POVRayParser.cpp:229-268: parsed, never stored */
         float p1[3], p2[3], r1, r2; rt_modifier mod; default_modifier(&mod);
         if (!parse_char(in, '{') || !parse_vector(in, p1) || !parse_word(in, ",")) return false;
         if (!in.number(r1)) return fail("cone radius expected");
         if (!parse_word(in, ",") || !parse_vector(in, p2) || !parse_word(in, ",")) return false;
         if (!in.number(r2)) return fail("cone radius expected");
         if (!parse_modifiers(in, &mod, out)) return false;
      } else if (first == "plane") {
         /* POVRayParser.cpp:270-299 */
         rt_plane p; default_modifier(&p.mod);
         if (!parse_char(in, '{') || !parse_vector(in, p.n) || !parse_word(in, ",")) return false;
         if (!in.number(p.d)) return fail("plane distance expected");
         bool ok = parse_modifiers(in, &p.mod, out);
         out.planes.push_back(p);
         if (!ok) return false;
      } else if (first == "triangle") {
         /* POVRayParser.cpp:374-430 */
         rt_triangle t; memset(&t, 0, sizeof t); default_modifier(&t.mod);
         if (!parse_char(in, '{') || !parse_vector(in, t.p1) || !parse_char(in, ',') || !parse_vector(in, t.p2) ||
             !parse_char(in, ',') || !parse_vector(in, t.p3) || !maybe_uv(in, t.vt1, t.vt2, t.vt3)) return false;
         bool ok = parse_modifiers(in, &t.mod, out);
         out.triangles.push_back(t);
         if (!ok) return false;
      } else if (first == "smooth_triangle") {
         /* POVRayParser.cpp:301-373: point, normal, point, normal, point, normal */
         rt_smooth_triangle t; memset(&t, 0, sizeof t); default_modifier(&t.mod);
         if (!parse_char(in, '{') || !parse_vector(in, t.p1) || !parse_char(in, ',') || !parse_vector(in, t.n1) ||
             !parse_char(in, ',') || !parse_vector(in, t.p2) || !parse_char(in, ',') || !parse_vector(in, t.n2) ||
             !parse_char(in, ',') || !parse_vector(in, t.p3) || !parse_char(in, ',') || !parse_vector(in, t.n3) ||
             !maybe_uv(in, t.vt1, t.vt2, t.vt3)) return false;
         bool ok = parse_modifiers(in, &t.mod, out);
         out.smooth.push_back(t);
         if (!ok) return false;
      } else {
         rt_set_error("parse error: Unrecognized object: %s", first.c_str());
         return false;
      }
      if (in.failed) return fail("line too long");
   }
   return true;
}

template <class T> const T *release(const std::vector<T> &v)
{
   T *p = (T *)malloc(sizeof(T) * (v.size() ? v.size() : 1));
   if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
   return p;      /* null when out of memory: rt_parse_pov checks */
}

} // namespace

extern "C" int rt_parse_pov(const char *path, rt_scene_desc **out)
{
   if (!path || !out) { rt_set_error("rt_parse_pov: null argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = NULL;
   FILE *f = fopen(path, "rb");
   if (!f) { rt_set_error("File \"%s\" not found", path); return RT_ERR_IO; }
   Scanner in;
   char chunk[1 << 16];
   size_t n;
   while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) in.data.append(chunk, n);
   fclose(f);
   SceneOut so;
   memset(&so.cam, 0, sizeof so.cam);
   if (!parse_file(in, so)) return RT_ERR_PARSE;
   rt_scene_desc *d = (rt_scene_desc *)calloc(1, sizeof(rt_scene_desc));
   if (!d) { rt_set_error("rt_parse_pov: out of memory"); return RT_ERR_IO; }
   d->camera = so.cam; d->has_camera = so.has_camera ? 1 : 0;
   d->n_spheres = (int32_t)so.spheres.size(); d->spheres = release(so.spheres);
   d->n_triangles = (int32_t)so.triangles.size(); d->triangles = release(so.triangles);
   d->n_boxes = (int32_t)so.boxes.size(); d->boxes = release(so.boxes);
   d->n_smooth_triangles = (int32_t)so.smooth.size(); d->smooth_triangles = release(so.smooth);
   d->n_planes = (int32_t)so.planes.size(); d->planes = release(so.planes);
   d->n_lights = (int32_t)so.lights.size(); d->lights = release(so.lights);
   d->n_textures = (int32_t)so.textures.size();
   d->textures = (rt_texture *)calloc(so.textures.size() ? so.textures.size() : 1, sizeof(rt_texture));
   if (!d->spheres || !d->triangles || !d->boxes || !d->smooth_triangles || !d->planes || !d->lights || !d->textures) {
      rt_scene_desc_free(d);
      rt_set_error("rt_parse_pov: out of memory");
      return RT_ERR_IO;
   }
   for (std::map<std::string, int>::const_iterator it = so.textures.begin(); it != so.textures.end(); ++it) {
      rt_texture &t = d->textures[it->second];
      /* the reference's sscanf("%s") keeps the closing quote in the name (POVRayParser.cpp:493); it is dropped here */
      std::string name = it->first;
      while (!name.empty() && (name[name.size() - 1] == '"' || name[name.size() - 1] == '}')) name.erase(name.size() - 1);
      snprintf(t.name, sizeof t.name, "%s", name.c_str());
   }
   *out = d;
   return RT_OK;
}

extern "C" int rt_load_bmp(const char *path, rt_texture *tex)
{
   if (!path || !tex) { rt_set_error("rt_load_bmp: null argument"); return RT_ERR_BAD_ARGUMENT; }
   FILE *f = fopen(path, "rb");
   if (!f) { rt_set_error("texture \"%s\" not found", path); return RT_ERR_IO; }
   unsigned char info[54];
   if (fread(info, 1, 54, f) != 54 || info[0] != 'B' || info[1] != 'M') { fclose(f); rt_set_error("%s is not a .bmp", path); return RT_ERR_PARSE; }
   int width, height;
   memcpy(&width, info + 18, 4); memcpy(&height, info + 22, 4);       /* kernel.cu:43-44 */
   int bits = info[28] | (info[29] << 8);
   if (width <= 0 || height <= 0 || width > 16384 || height > 16384 || bits != 24) { fclose(f); rt_set_error("%s: only 24-bit bottom-up images are supported", path); return RT_ERR_UNSUPPORTED; }
   size_t n = (size_t)width * (size_t)height;
   std::vector<unsigned char> data(3 * n);
   if (fread(data.data(), 1, 3 * n, f) != 3 * n) { fclose(f); rt_set_error("%s is truncated", path); return RT_ERR_PARSE; }
   fclose(f);
   uint8_t *out = (uint8_t *)malloc(4 * n);
   if (!out) { rt_set_error("rt_load_bmp: out of memory"); return RT_ERR_IO; }
   for (size_t i = 0; i < n; i++) {                                    /* kernel.cu:53-59: BGR -> RGB0, rows as stored */
      out[4 * i] = data[3 * i + 2]; out[4 * i + 1] = data[3 * i + 1]; out[4 * i + 2] = data[3 * i]; out[4 * i + 3] = 0;
   }
   tex->width = width; tex->height = height; tex->rgba = out;      /* whatever tex->rgba pointed to before stays the caller's */
   return RT_OK;
}

extern "C" void rt_scene_desc_free(rt_scene_desc *d)
{
   if (!d) return;
   free((void *)d->spheres); free((void *)d->triangles); free((void *)d->boxes);
   free((void *)d->smooth_triangles); free((void *)d->planes); free((void *)d->lights);
   if (d->textures) { for (int i = 0; i < d->n_textures; i++) free(d->textures[i].rgba); free(d->textures); }
   free(d);
}
