/*
 * rt_scene_build.cpp -- host side of scene set-up: turns a scene description
 * into the flat device records of rt_common.h.
 *
 * Replaces, on the host, what the reference does with device-side `new` and a
 * GPU merge sort: initScene (kernel.cu:384-464) and formBVH (bvh.cpp:30-223).
 * The resulting TREE IS THE REFERENCE'S TREE -- same leaf order, same pairing,
 * same boxes bit for bit -- because which primitive wins a near-tie depends on
 * the order the traversal meets them (kernel.cu:108-134).
 *
 * Arithmetic follows GLM 0.9.3's operation order (mat4*vec4 and vec4*mat4:
 * glm/core/type_mat4x4.inl:689-714; cross/normalize: func_geometric.inl:199-248)
 * in IEEE binary32 without contraction.
 */
#include "rt_common.h"

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string>
#include <algorithm>
#include <new>
#include <thread>
#include <memory>
#include <vector>

namespace {

const float kEps = 0.0001f;   /* EPSILON, Util.h:4 */

struct F3 { float x, y, z; };
inline F3 f3(const float *p) { return F3{p[0], p[1], p[2]}; }
inline F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 cross(F3 x, F3 y) { return F3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline F3 normalize(F3 v)
{
   float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
   float inv = 1.0f / std::sqrt(sqr);
   return F3{v.x * inv, v.y * inv, v.z * inv};
}
inline float lo(float a, float b) { return a < b ? a : b; }   /* mymin, Util.h:36-38 */
inline float hi(float a, float b) { return a > b ? a : b; }   /* mymax, Util.h:40-42 */

/* M * (x,y,z,w), column-major M; only .xyz is ever kept by the callers */
inline F3 mat_vec(const float *m, float x, float y, float z, float w)
{
   return F3{m[0] * x + m[4] * y + m[8] * z + m[12] * w,
             m[1] * x + m[5] * y + m[9] * z + m[13] * w,
             m[2] * x + m[6] * y + m[10] * z + m[14] * w};
}
/* (x,y,z,w) * M: component c is the dot with column c */
inline F3 vec_mat(float x, float y, float z, float w, const float *m)
{
   return F3{m[0] * x + m[1] * y + m[2] * z + m[3] * w,
             m[4] * x + m[5] * y + m[6] * z + m[7] * w,
             m[8] * x + m[9] * y + m[10] * z + m[11] * w};
}

struct Box3 { F3 mn, mx; };
/* every BoundingBox construction pads by EPSILON (BoundingBox.h:11-15) */
inline Box3 padded(F3 mn, F3 mx)
{
   return Box3{F3{mn.x - kEps, mn.y - kEps, mn.z - kEps}, F3{mx.x + kEps, mx.y + kEps, mx.z + kEps}};
}
/* generateBoundingBox, BoundingBox.h:24-46 */
Box3 corners_box(F3 mn, F3 mx, const float *trans)
{
   float lo3[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi3[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
   for (int i = 0; i < 8; i++) {
      F3 p = mat_vec(trans, (i & 1) ? mx.x : mn.x, (i & 2) ? mx.y : mn.y, (i & 4) ? mx.z : mn.z, 1.0f);
      float q[3] = {p.x, p.y, p.z};
      for (int a = 0; a < 3; a++) {
         if (q[a] < lo3[a]) lo3[a] = q[a];
         if (q[a] > hi3[a]) hi3[a] = q[a];
      }
   }
   return padded(f3(lo3), f3(hi3));
}
/* combineBoundingBox, BoundingBox.h:48-52 */
inline Box3 merged(const Box3 &a, const Box3 &b)
{
   return padded(F3{lo(a.mn.x, b.mn.x), lo(a.mn.y, b.mn.y), lo(a.mn.z, b.mn.z)},
                 F3{hi(a.mx.x, b.mx.x), hi(a.mx.y, b.mx.y), hi(a.mx.z, b.mx.z)});
}

bool is_identity(const float *m)
{
   for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++)
         if (m[c * 4 + r] != (c == r ? 1.0f : 0.0f)) return false;
   return true;
}

void put(float *dst, F3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

void fill_xform(rt_dev_xform *x, const float *inv)
{
   for (int k = 0; k < 3; k++)
      for (int c = 0; c < 4; c++) x->r[k][c] = inv[c * 4 + k];
}

/* a material as a key: its bytes (a scene may give every primitive a material of its own: a million keys) */
struct MatKey {
   rt_material m;
   bool operator==(const MatKey &o) const { return memcmp(&m, &o.m, sizeof m) == 0; }
};
struct MatKeyHash {
   size_t operator()(const MatKey &k) const
   {
      uint64_t h = 0xcbf29ce484222325ULL;
      uint32_t w[(sizeof(rt_material) + 3) / 4] = {0};
      memcpy(w, &k.m, sizeof k.m);
      for (uint32_t v : w) { h ^= v; h *= 0x100000001b3ULL; h ^= h >> 29; }
      return (size_t)h;
   }
};

/* the deduplicated materials, in order of first appearance */
struct MatTable {
   std::vector<rt_material> materials;
   std::unordered_map<MatKey, uint32_t, MatKeyHash> index;
   MatKey last_key;
   uint32_t last_id = 0;
   bool have_last = false;
   uint32_t id_of(const rt_material &m)
   {
      MatKey k; memset(&k, 0, sizeof k); k.m = m;
      /* (a mesh's triangles share one material: the one asked for last is the likely answer) */
      if (have_last && k == last_key) return last_id;
      uint32_t id;
      auto it = index.find(k);
      if (it != index.end()) id = it->second;
      else {
         id = (uint32_t)materials.size();
         materials.push_back(m);
         index[k] = id;
      }
      last_key = k; last_id = id; have_last = true;
      return id;
   }
};

/* The per-primitive records, written by index (several threads fill disjoint ranges of a large scene: a million
 * records are a gigabyte of host memory touched for the first time, which one thread does no faster than the memory
 * is mapped for it).  prims, xforms and shades are handed to the caller as they are. */
struct Builder {
   rt_dev_prim *prims = nullptr;
   rt_dev_xform *xforms = nullptr;
   rt_dev_shade *shades = nullptr;
   std::vector<Box3> boxes;       /* getBoundingBox() per primitive */
   std::vector<F3> centers;       /* getCenter() per primitive: the sort key */
   MatTable mats;                 /* the scene's table: the planes' materials first, then the primitives' */

   ~Builder() { free(prims); free(xforms); free(shades); }
   bool allocate(size_t n)
   {
      const size_t m = n ? n : 1;
      prims = (rt_dev_prim *)malloc(m * sizeof(rt_dev_prim));
      xforms = (rt_dev_xform *)malloc(m * sizeof(rt_dev_xform));
      shades = (rt_dev_shade *)malloc(m * sizeof(rt_dev_shade));
      if (!prims || !xforms || !shades) return false;
      boxes.resize(n); centers.resize(n);
      return true;
   }

   /* (the material number is the filling thread's own until the tables are merged) */
   void common(size_t at, uint32_t type, const rt_modifier &mod, MatTable &mt, int &any_xform)
   {
      rt_dev_prim &p = prims[at];
      memset(&p, 0, sizeof p);
      p.type = type;
      p.material = mt.id_of(mod.mat);
      if (!is_identity(mod.inv_trans)) { p.flags |= RT_FLAG_XFORM; any_xform = 1; }
      fill_xform(&xforms[at], mod.inv_trans);
   }

   /* Sphere.h:10-15, 26-32 */
   void add_sphere(size_t at, const rt_sphere &s, MatTable &mt, int &any_xform)
   {
      common(at, RT_PRIM_SPHERE, s.mod, mt, any_xform);
      rt_dev_prim &p = prims[at];
      p.g[0] = s.c[0]; p.g[1] = s.c[1]; p.g[2] = s.c[2]; p.g[3] = s.r;
      rt_dev_shade &sh = shades[at]; memset(&sh, 0, sizeof sh);
      F3 wc = mat_vec(s.mod.trans, s.c[0], s.c[1], s.c[2], 1.0f);
      put(sh.s, wc);
      centers[at] = wc;
      boxes[at] = corners_box(F3{s.c[0] - s.r, s.c[1] - s.r, s.c[2] - s.r},
                              F3{s.c[0] + s.r, s.c[1] + s.r, s.c[2] + s.r}, s.mod.trans);
   }

   /* Triangle.h:11-42; SmoothTriangle.h:12-28 */
   void add_triangle(size_t at, const float *a, const float *b, const float *c, const rt_modifier &mod,
                     const float *n1, const float *n2, const float *n3, const float *vt1, const float *vt2, const float *vt3,
                     MatTable &mt, int &any_xform)
   {
      common(at, n1 ? RT_PRIM_SMOOTH : RT_PRIM_TRIANGLE, mod, mt, any_xform);
      rt_dev_prim &p = prims[at];
      F3 p1 = f3(a), p2 = f3(b), p3 = f3(c);
      F3 n = normalize(cross(p2 - p1, p3 - p1));
      put(&p.g[0], p1); put(&p.g[3], p2); put(&p.g[6], p3); put(&p.g[9], n);
      rt_dev_shade &sh = shades[at]; memset(&sh, 0, sizeof sh);
      if (n1) {
         put(&sh.s[0], vec_mat(n1[0], n1[1], n1[2], 0.0f, mod.inv_trans));
         put(&sh.s[3], vec_mat(n2[0], n2[1], n2[2], 0.0f, mod.inv_trans));
         put(&sh.s[6], vec_mat(n3[0], n3[1], n3[2], 0.0f, mod.inv_trans));
      } else {
         put(&sh.s[0], vec_mat(n.x, n.y, n.z, 0.0f, mod.inv_trans));
      }
      sh.s[9] = vt1[0]; sh.s[10] = vt1[1]; sh.s[11] = vt2[0]; sh.uv[0] = vt2[1]; sh.uv[1] = vt3[0]; sh.uv[2] = vt3[1];
      /* sort key: centroid as a ROW vector times trans -- the translation column never enters (Triangle.h:28-31) */
      F3 s3 = (p1 + p2) + p3;
      F3 cen = F3{s3.x / 3.0f, s3.y / 3.0f, s3.z / 3.0f};
      centers[at] = vec_mat(cen.x, cen.y, cen.z, 1.0f, mod.trans);
      F3 t1 = mat_vec(mod.trans, p1.x, p1.y, p1.z, 1.0f);
      F3 t2 = mat_vec(mod.trans, p2.x, p2.y, p2.z, 1.0f);
      F3 t3 = mat_vec(mod.trans, p3.x, p3.y, p3.z, 1.0f);
      F3 mn = F3{lo(lo(t1.x, t2.x), t3.x), lo(lo(t1.y, t2.y), t3.y), lo(lo(t1.z, t2.z), t3.z)};
      F3 mx = F3{hi(hi(t1.x, t2.x), t3.x), hi(hi(t1.y, t2.y), t3.y), hi(hi(t1.z, t2.z), t3.z)};
      boxes[at] = padded(mn, mx);
   }

   /* Box.h:13-34 */
   void add_box(size_t at, const rt_box &b, MatTable &mt, int &any_xform)
   {
      common(at, RT_PRIM_BOX, b.mod, mt, any_xform);
      rt_dev_prim &p = prims[at];
      F3 mn = F3{lo(b.p1[0], b.p2[0]), lo(b.p1[1], b.p2[1]), lo(b.p1[2], b.p2[2])};
      F3 mx = F3{hi(b.p1[0], b.p2[0]), hi(b.p1[1], b.p2[1]), hi(b.p1[2], b.p2[2])};
      put(&p.g[0], mn); put(&p.g[3], mx);
      rt_dev_shade &sh = shades[at]; memset(&sh, 0, sizeof sh);
      for (int c = 0; c < 3; c++)
         for (int r = 0; r < 3; r++) sh.s[c * 3 + r] = b.mod.trans[c * 4 + r];
      F3 s2 = mn + mx;
      F3 mid = F3{s2.x / 2.0f, s2.y / 2.0f, s2.z / 2.0f};
      centers[at] = mat_vec(b.mod.trans, mid.x, mid.y, mid.z, 1.0f);
      boxes[at] = corners_box(mn, mx, b.mod.trans);
   }
};

inline float axis_of(const F3 &v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; }

/* One sort round of formBVH: every queued range is sorted on `axis` by the
 * bottom-up merge of bvh.cpp:106-171.  In a merge the LEFT run's head is taken
 * only when its key is strictly smaller, so equal keys come out right-run
 * first -- this is not std::stable_sort and must not be replaced by it. */
void merge_sort_range_literal(int32_t *ids, int32_t *tmp, int n, const std::vector<F3> &centers, int axis);

/* What that merge computes, for keys without a NaN among them: each run it has built is in order of (key ascending,
 * place in the input descending) -- runs of one trivially, and a merge of a left and a right run that takes the right
 * head on equal keys keeps it so, the right run holding the later places.  So the outcome is the input sorted by key
 * with equal keys in reverse input order, whatever the run boundaries were, and a stable radix sort of the input fed
 * back to front gives the same permutation.  (-0 and +0 are equal keys to `<`: both are keyed as +0.)  Long ranges are
 * sorted that way -- the merge's passes with a dependent load per comparison were most of a scene's host build -- short
 * ones and any range with a NaN key by the merge itself; tests/native/bvh_sort_equivalence.cpp runs both on random,
 * heavily tied, signed-zero and infinite keys. */
void merge_sort_range(int32_t *ids, int32_t *tmp, int n, const std::vector<F3> &centers, int axis)
{
   if (n >= 192) {
      static thread_local std::vector<uint64_t> a, b;
      a.resize((size_t)n); b.resize((size_t)n);
      bool ok = true;
      uint32_t all_or = 0, all_and = 0xffffffffu;
      for (int i = 0; i < n && ok; i++) {
         const int32_t id = ids[n - 1 - i];
         const float k = axis_of(centers[(size_t)id], axis) + 0.0f;
         ok = k == k;
         uint32_t u; memcpy(&u, &k, 4);
         u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      /* unsigned order = float order */
         all_or |= u; all_and &= u;
         a[(size_t)i] = ((uint64_t)u << 32) | (uint32_t)id;
      }
      if (ok) {
         uint64_t *src = a.data(), *dst = b.data();
         for (int shift = 32; shift < 64; shift += 8) {
            if ((((all_or ^ all_and) >> (shift - 32)) & 0xffu) == 0) continue;      /* every key has the same digit here */
            size_t count[257] = {0};
            for (int i = 0; i < n; i++) count[((src[i] >> shift) & 0xffu) + 1]++;
            for (int d = 0; d < 256; d++) count[d + 1] += count[d];
            for (int i = 0; i < n; i++) dst[count[(src[i] >> shift) & 0xffu]++] = src[i];
            uint64_t *t = src; src = dst; dst = t;
         }
         for (int i = 0; i < n; i++) ids[i] = (int32_t)(uint32_t)src[i];
         return;
      }
   }
   merge_sort_range_literal(ids, tmp, n, centers, axis);
}

void merge_sort_range_literal(int32_t *ids, int32_t *tmp, int n, const std::vector<F3> &centers, int axis)
{
   int32_t *src = ids, *dst = tmp;
   for (int width = 1; width < n; width *= 2) {
      for (int base = 0; base < n; base += 2 * width) {
         int i1 = base, i2 = base + width;
         int e1 = i2 < n ? i2 : n;
         int e2 = (i2 + width) < n ? (i2 + width) : n;
         for (int i = base; i < e2; i++) {
            if (i1 < e1 && (i2 >= e2 || axis_of(centers[src[i1]], axis) < axis_of(centers[src[i2]], axis)))
               dst[i] = src[i1++];
            else
               dst[i] = src[i2++];
         }
      }
      int32_t *t = src; src = dst; dst = t;
   }
   if (src != ids) memcpy(ids, src, sizeof(int32_t) * (size_t)n);
}

struct TreeNode {      /* BVHNode, bvh.h:10-19 */
   int left, right;    /* indices into the TreeNode pool */
   int prim;           /* >= 0: leaf */
   Box3 bb;
};

/* Pre-order numbering (a parent is directly followed by its left subtree, so the
 * first child visited is often in the next line), with an explicit stack:
 * first number the internal nodes, then fill the records. */
void flatten_tree(const std::vector<TreeNode> &pool, int root, std::vector<rt_bvh_node> &out, int *depth_out)
{
   std::vector<int32_t> index(pool.size(), -1);
   std::vector<std::pair<int, int> > stack;   /* node, level */
   int32_t next = 0;
   int depth = 0;
   stack.push_back(std::make_pair(root, 1));
   std::vector<int> order;
   while (!stack.empty()) {
      std::pair<int, int> it = stack.back(); stack.pop_back();
      const TreeNode &n = pool[it.first];
      index[it.first] = next++;
      order.push_back(it.first);
      if (it.second > depth) depth = it.second;
      if (pool[n.right].prim < 0) stack.push_back(std::make_pair(n.right, it.second + 1));
      if (pool[n.left].prim < 0) stack.push_back(std::make_pair(n.left, it.second + 1));
   }
   out.resize((size_t)next);
   for (size_t k = 0; k < order.size(); k++) {
      const TreeNode &n = pool[order[k]];
      const TreeNode &l = pool[n.left];
      const TreeNode &r = pool[n.right];
      rt_bvh_node rec; memset(&rec, 0, sizeof rec);
      put(rec.lmin, l.bb.mn); put(rec.lmax, l.bb.mx);
      put(rec.rmin, r.bb.mn); put(rec.rmax, r.bb.mx);
      rec.left = l.prim >= 0 ? ~l.prim : index[n.left];
      rec.right = r.prim >= 0 ? ~r.prim : index[n.right];
      out[(size_t)index[order[k]]] = rec;
   }
   *depth_out = depth;
}

/* ---- opt-in tree: binned surface-area heuristic ---------------------------
 * NOT the reference's tree.  Same node layout, same leaves (one primitive each), so the traversal kernel is
 * unchanged, but the visiting order differs and with it the outcome of near-ties (kernel.cu:108-122): frames
 * made with it are reported apart from the parity numbers.  Boxes are plain unions of the (already padded)
 * primitive boxes.  Depth is capped at ceil(log2 n) + 8 by falling back to median splits, because the
 * traversal stack lives in LDS. */
inline Box3 unite(const Box3 &a, const Box3 &b)
{
   return Box3{F3{lo(a.mn.x, b.mn.x), lo(a.mn.y, b.mn.y), lo(a.mn.z, b.mn.z)}, F3{hi(a.mx.x, b.mx.x), hi(a.mx.y, b.mx.y), hi(a.mx.z, b.mx.z)}};
}
inline float half_area(const Box3 &b)
{
   float dx = b.mx.x - b.mn.x, dy = b.mx.y - b.mn.y, dz = b.mx.z - b.mn.z;
   return dx * dy + dy * dz + dz * dx;
}
inline F3 box_center(const Box3 &b) { return F3{0.5f * (b.mn.x + b.mx.x), 0.5f * (b.mn.y + b.mx.y), 0.5f * (b.mn.z + b.mx.z)}; }

struct SahBuilder {
   const std::vector<Box3> &boxes;
   const std::vector<F3> &cent;       /* box_center of every primitive's box */
   std::vector<int32_t> &ids;
   std::vector<TreeNode> &pool;
   enum { BINS = 16 };

   static int bin_of(float c, float lo_c, float scale)
   {
      int k = (int)((c - lo_c) * scale);
      return k >= BINS ? BINS - 1 : (k < 0 ? 0 : k);
   }

   /* `par`: how many more levels hand their right half to another thread (large ranges only).  The halves work on
    * disjoint ranges of `ids` and on node pools of their own; the tree is the same whatever `par` is. */
   int build(int first, int last, int depth_left, int par = 0)     /* [first, last) */
   {
      int n = last - first;
      if (n == 1) { int p = ids[(size_t)first]; pool.push_back(TreeNode{-1, -1, p, boxes[(size_t)p]}); return (int)pool.size() - 1; }
      F3 cmn = cent[(size_t)ids[(size_t)first]], cmx = cmn;
      for (int i = first + 1; i < last; i++) {
         const F3 &c = cent[(size_t)ids[(size_t)i]];
         cmn = F3{lo(cmn.x, c.x), lo(cmn.y, c.y), lo(cmn.z, c.z)}; cmx = F3{hi(cmx.x, c.x), hi(cmx.y, c.y), hi(cmx.z, c.z)};
      }
      int levels_needed = 0; while ((1 << levels_needed) < n) levels_needed++;
      int mid = -1;
      /* (a handful of primitives is split at the median: the bins would say little more and most nodes are down here) */
      if (depth_left > levels_needed && n > 4) {
         /* one pass over the range fills the bins of all three axes */
         const float lo3[3] = {cmn.x, cmn.y, cmn.z}, ext3[3] = {cmx.x - cmn.x, cmx.y - cmn.y, cmx.z - cmn.z};
         float scale3[3];
         for (int a = 0; a < 3; a++) scale3[a] = ext3[a] > 0.0f ? (float)BINS / ext3[a] : 0.0f;
         const Box3 none{F3{FLT_MAX, FLT_MAX, FLT_MAX}, F3{-FLT_MAX, -FLT_MAX, -FLT_MAX}};
         Box3 bb[3][BINS]; int cnt[3][BINS];
         for (int a = 0; a < 3; a++) for (int k = 0; k < BINS; k++) { bb[a][k] = none; cnt[a][k] = 0; }
         for (int i = first; i < last; i++) {
            const int32_t p = ids[(size_t)i];
            const Box3 &pb = boxes[(size_t)p];
            const F3 &c = cent[(size_t)p];
            const float c3[3] = {c.x, c.y, c.z};
            for (int a = 0; a < 3; a++) {
               const int k = bin_of(c3[a], lo3[a], scale3[a]);
               bb[a][k] = unite(bb[a][k], pb); cnt[a][k]++;
            }
         }
         float best = FLT_MAX; int best_axis = -1, best_bin = -1;
         for (int axis = 0; axis < 3; axis++) {
            if (!(ext3[axis] > 0.0f) || !(scale3[axis] < FLT_MAX)) continue;
            float right_area[BINS]; int right_cnt[BINS];
            Box3 acc = none; int c = 0;
            for (int k = BINS - 1; k > 0; k--) {
               if (cnt[axis][k]) { acc = unite(acc, bb[axis][k]); c += cnt[axis][k]; }
               right_area[k] = c ? half_area(acc) : 0.0f; right_cnt[k] = c;
            }
            acc = none; c = 0;
            for (int k = 0; k < BINS - 1; k++) {
               if (cnt[axis][k]) { acc = unite(acc, bb[axis][k]); c += cnt[axis][k]; }
               if (c == 0 || right_cnt[k + 1] == 0) continue;
               float cost = half_area(acc) * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
               if (cost < best) { best = cost; best_axis = axis; best_bin = k; }
            }
         }
         if (best_axis >= 0) {
            const float lo_c = lo3[best_axis], scale = scale3[best_axis];
            int i = first, j = last - 1;
            while (i <= j) {
               const int k = bin_of(axis_of(cent[(size_t)ids[(size_t)i]], best_axis), lo_c, scale);
               if (k <= best_bin) i++; else { std::swap(ids[(size_t)i], ids[(size_t)j]); j--; }
            }
            if (i > first && i < last) mid = i;
         }
      }
      if (mid < 0) {
         /* median split on the widest centroid axis (also the forced choice once the depth budget is tight) */
         int axis = 0;
         float ex = cmx.x - cmn.x, ey = cmx.y - cmn.y, ez = cmx.z - cmn.z;
         if (ey > ex && ey >= ez) axis = 1; else if (ez > ex && ez > ey) axis = 2;
         mid = first + n / 2;
         std::nth_element(ids.begin() + first, ids.begin() + mid, ids.begin() + last, [&](int32_t a, int32_t b) {
            float ca = axis_of(cent[(size_t)a], axis), cb2 = axis_of(cent[(size_t)b], axis);
            return ca < cb2 || (ca == cb2 && a < b);
         });
      }
      int l, r;
      if (par > 0 && n >= 2048) {
         std::vector<TreeNode> other_pool;
         other_pool.reserve((size_t)(2 * (last - mid)));
         SahBuilder other{boxes, cent, ids, other_pool};
         int r_local = -1;
         bool threaded = true;
         std::thread worker;
         bool worker_failed = false;
         try {
            worker = std::thread([&]() { try { r_local = other.build(mid, last, depth_left - 1, par - 1); } catch (...) { worker_failed = true; } });
         } catch (...) { threaded = false; }
         /* (whatever this half throws, the other half's thread is joined before the exception goes on) */
         try { l = build(first, mid, depth_left - 1, par - 1); } catch (...) { if (threaded) worker.join(); throw; }
         if (threaded) worker.join(); else r_local = other.build(mid, last, depth_left - 1, 0);
         if (worker_failed) throw std::bad_alloc();
         const int off = (int)pool.size();
         for (TreeNode nd : other_pool) {
            if (nd.prim < 0) { nd.left += off; nd.right += off; }
            pool.push_back(nd);
         }
         r = r_local + off;
      } else {
         l = build(first, mid, depth_left - 1, 0);
         r = build(mid, last, depth_left - 1, 0);
      }
      pool.push_back(TreeNode{l, r, -1, unite(pool[(size_t)l].bb, pool[(size_t)r].bb)});
      return (int)pool.size() - 1;
   }
};

/* The second tree over `n` primitive boxes, unwidened (root 0); empty if it cannot be had (one primitive, or out of
 * memory: the scene goes without). */
void build_filter_tree(const std::vector<Box3> &boxes, const std::vector<F3> &centers, int n, std::vector<rt_bvh_node> &filter, int &filter_depth)
{
   try {
      std::vector<int32_t> ids((size_t)n);
      for (int i = 0; i < n; i++) ids[(size_t)i] = i;
      std::vector<TreeNode> pool;
      pool.reserve((size_t)(2 * n));
      int levels = 0; while ((1 << levels) < n) levels++;
      SahBuilder sb{boxes, centers, ids, pool};
      /* no deeper than the traversal stack's share of LDS where the primitive count allows that: a tree that pushed
       * the stack into global memory would cost more than its better splits win */
      int budget = levels + 2 > RT_FILTER_DEPTH ? levels + 2 : RT_FILTER_DEPTH;
      int root_node = sb.build(0, n, budget, 3);
      if (pool[(size_t)root_node].prim < 0) flatten_tree(pool, root_node, filter, &filter_depth);
   } catch (...) { filter.clear(); filter_depth = 0; }
}

/* Every box of the second tree contains the reference's own (padded) boxes of the primitives below it.  It is
 * widened by 1e-4 plus a few units in the last place of its coordinates, far more than the rounding of the
 * box test, so that a ray the reference's test lets into a primitive's own box is never turned away there; and by
 * the margin for the rays' origins.  A tree with a bound out of the box test's range is dropped. */
void widen_filter_tree(std::vector<rt_bvh_node> &filter, int &filter_depth, float origin_pad)
{
   for (size_t k = 0; k < filter.size(); k++) {
      float *v = filter[k].lmin;      /* lmin, lmax, rmin, rmax */
      for (int j = 0; j < 12; j++) {
         const float pad = 1e-4f + 8e-6f * std::fabs(v[j]) + origin_pad;
         const bool is_max = (j / 3) & 1;
         v[j] = is_max ? v[j] + pad : v[j] - pad;
      }
   }
   for (size_t k = 0; k < filter.size(); k++) {
      const float *v = filter[k].lmin;
      bool bad = false;
      for (int j = 0; j < 12; j++) if (!(std::fabs(v[j]) <= 1.099511627776e12f)) bad = true;
      if (bad) { filter.clear(); filter_depth = 0; break; }
   }
}

/* A second tree still being built when rt_flatten_scene_ex returns (RT_FLATTEN_DEFER_FILTER): the thread, what it
 * reads and what it writes.  rt_flat_scene_finish joins it. */
struct PendingFilter {
   std::vector<Box3> boxes;
   std::vector<F3> centers;
   int n = 0;
   float origin_pad = 0.0f;
   std::vector<rt_bvh_node> nodes;
   int depth = 0;
   std::thread worker;
   ~PendingFilter() { if (worker.joinable()) worker.join(); }
};

/* a material's texture reference must name a texture whose pixels are there */
/* the same test without the message (what the filling threads use) */
bool tex_known(const rt_scene_desc *d, int tex_id)
{
   return tex_id == -1 || !(tex_id < 0 || tex_id >= d->n_textures || !d->textures || !d->textures[tex_id].rgba ||
                            d->textures[tex_id].width <= 0 || d->textures[tex_id].height <= 0);
}
bool tex_ok(const rt_scene_desc *d, int tex_id)
{
   if (tex_id == -1) return true;
   if (tex_id < 0 || tex_id >= d->n_textures || !d->textures || !d->textures[tex_id].rgba ||
       d->textures[tex_id].width <= 0 || d->textures[tex_id].height <= 0) {
      rt_set_error("material refers to texture %d, which is not loaded (rt_load_bmp)", tex_id);
      return false;
   }
   return true;
}

template <class T> T *copy_out(const std::vector<T> &v)
{
   T *p = (T *)malloc(sizeof(T) * (v.size() ? v.size() : 1));
   if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
   return p;
}

} // namespace

static thread_local char g_error[512];

extern "C" void rt_set_error(const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_error, sizeof g_error, fmt, ap);
   va_end(ap);
}
extern "C" const char *rt_last_error(void) { return g_error; }

extern "C" int rt_flatten_scene(const rt_scene_desc *d, rt_flat_scene *out) { return rt_flatten_scene_ex(d, 0, out); }

extern "C" int rt_flatten_scene_ex(const rt_scene_desc *d, int flags, rt_flat_scene *out)
{
   if (!d || !out) { rt_set_error("rt_flatten_scene: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (d->n_spheres < 0 || d->n_triangles < 0 || d->n_boxes < 0 || d->n_smooth_triangles < 0 ||
       d->n_planes < 0 || d->n_lights < 0) { rt_set_error("negative primitive count"); return RT_ERR_BAD_ARGUMENT; }
   if ((d->n_spheres && !d->spheres) || (d->n_triangles && !d->triangles) || (d->n_boxes && !d->boxes) ||
       (d->n_smooth_triangles && !d->smooth_triangles) || (d->n_planes && !d->planes) || (d->n_lights && !d->lights)) {
      rt_set_error("primitive count without array"); return RT_ERR_BAD_ARGUMENT;
   }
   long long total = (long long)d->n_spheres + d->n_triangles + d->n_boxes + d->n_smooth_triangles;
   if (total > 0x3fffffffLL) { rt_set_error("too many primitives"); return RT_ERR_TOO_LARGE; }
   memset(out, 0, sizeof *out);

   Builder b;
   if (!b.allocate((size_t)total)) { rt_set_error("out of memory for %lld primitive records", total); return RT_ERR_BAD_ARGUMENT; }

   /* planes first, as initScene does (kernel.cu:411-416); they stay out of the tree */
   std::vector<rt_dev_plane> planes((size_t)d->n_planes);
   for (int i = 0; i < d->n_planes; i++) {
      const rt_plane &p = d->planes[i];
      if (!tex_ok(d, p.mod.mat.tex_id)) return RT_ERR_BAD_ARGUMENT;
      rt_dev_plane &o = planes[(size_t)i];
      memset(&o, 0, sizeof o);
      o.n[0] = p.n[0]; o.n[1] = p.n[1]; o.n[2] = p.n[2]; o.d = p.d;
      fill_xform(&o.inv, p.mod.inv_trans);
      put(o.wsN, mat_vec(p.mod.trans, p.n[0], p.n[1], p.n[2], 0.0f));
      if (!is_identity(p.mod.inv_trans)) o.flags |= RT_FLAG_XFORM;
      o.material = b.mats.id_of(p.mod.mat);
   }
   /* geomList order: spheres, triangles, boxes, smooth triangles (kernel.cu:419-456).  Ranges of that order are filled
    * side by side; each range numbers its materials by first appearance within it, and the ranges' tables are merged in
    * order afterwards, which gives the numbers one pass over everything would have given. */
   int any_xform = 0;
   {
      const long long first_tri = d->n_spheres, first_box = first_tri + d->n_triangles, first_smooth = first_box + d->n_boxes;
      /* -1: filled; otherwise the first primitive of the range whose texture reference is bad */
      auto fill = [&](long long lo_i, long long hi_i, MatTable &mt, int &xf) -> long long {
         for (long long g = lo_i; g < hi_i; g++) {
            if (g < first_tri) {
               const rt_sphere &sp = d->spheres[g];
               if (!tex_known(d, sp.mod.mat.tex_id)) return g;
               b.add_sphere((size_t)g, sp, mt, xf);
            } else if (g < first_box) {
               const rt_triangle &t = d->triangles[g - first_tri];
               if (!tex_known(d, t.mod.mat.tex_id)) return g;
               b.add_triangle((size_t)g, t.p1, t.p2, t.p3, t.mod, NULL, NULL, NULL, t.vt1, t.vt2, t.vt3, mt, xf);
            } else if (g < first_smooth) {
               const rt_box &bx = d->boxes[g - first_box];
               if (!tex_known(d, bx.mod.mat.tex_id)) return g;
               b.add_box((size_t)g, bx, mt, xf);
            } else {
               const rt_smooth_triangle &t = d->smooth_triangles[g - first_smooth];
               if (!tex_known(d, t.mod.mat.tex_id)) return g;
               b.add_triangle((size_t)g, t.p1, t.p2, t.p3, t.mod, t.n1, t.n2, t.n3, t.vt1, t.vt2, t.vt3, mt, xf);
            }
         }
         return -1;
      };
      unsigned hw = std::thread::hardware_concurrency();
      int parts = total >= 65536 ? (int)(hw > 8 ? 8 : (hw > 1 ? hw : 1)) : 1;
      std::vector<MatTable> tables((size_t)parts);
      std::vector<int> xf((size_t)parts, 0);
      std::vector<long long> bad((size_t)parts, -1);
      auto range_lo = [&](int k) { return total * k / parts; };
      {
         std::vector<std::thread> workers;
         struct JoinAll { std::vector<std::thread> &w; ~JoinAll() { for (auto &t : w) if (t.joinable()) t.join(); } } join_all{workers};
         for (int k = 1; k < parts; k++) {
            try { workers.emplace_back([&, k]() { bad[(size_t)k] = fill(range_lo(k), range_lo(k + 1), tables[(size_t)k], xf[(size_t)k]); }); }
            catch (...) { bad[(size_t)k] = fill(range_lo(k), range_lo(k + 1), tables[(size_t)k], xf[(size_t)k]); }      /* no thread to be had: here, then */
         }
         bad[0] = fill(range_lo(0), range_lo(1), tables[0], xf[0]);
      }
      for (int k = 0; k < parts; k++) {
         if (bad[(size_t)k] >= 0) {
            /* the same check again, for its message */
            const long long g = bad[(size_t)k];
            const int tex = g < first_tri ? d->spheres[g].mod.mat.tex_id : g < first_box ? d->triangles[g - first_tri].mod.mat.tex_id
                          : g < first_smooth ? d->boxes[g - first_box].mod.mat.tex_id : d->smooth_triangles[g - first_smooth].mod.mat.tex_id;
            (void)tex_ok(d, tex);
            return RT_ERR_BAD_ARGUMENT;
         }
      }
      {
         size_t most = b.mats.materials.size();
         for (int k = 0; k < parts; k++) most += tables[(size_t)k].materials.size();
         b.mats.materials.reserve(most); b.mats.index.reserve(most);
      }
      for (int k = 0; k < parts; k++) {
         const MatTable &mt = tables[(size_t)k];
         std::vector<uint32_t> global(mt.materials.size());
         for (size_t m = 0; m < mt.materials.size(); m++) global[m] = b.mats.id_of(mt.materials[m]);
         for (long long g = range_lo(k); g < range_lo(k + 1); g++) b.prims[g].material = global[b.prims[g].material];
         any_xform |= xf[(size_t)k];
      }
   }

   int n = (int)total;
   std::vector<int32_t> order((size_t)(n > 0 ? n : 1)), tmp((size_t)(n > 0 ? n : 1));
   for (int i = 0; i < n; i++) order[(size_t)i] = i;

   /* The second tree: beside the reference's tree, a surface-area tree over the same primitive boxes, every box
    * widened a little.  It decides nothing the reference decides; the traversal kernel uses it to prove what the
    * reference's walk returns (no primitive met; or this one primitive, with no rival near it), and walks the reference's
    * tree for every ray it cannot prove that for.  Built only beside the reference's tree, and on a thread of its own
    * while that one is built here: the two read the same boxes and share nothing else. */
   /* (the centres of the primitives' boxes, for the surface-area builder: both of its uses read them) */
   std::vector<F3> box_centers;
   if (n > 0 && (flags & RT_SCENE_FAST_BVH || !(flags & (RT_SCENE_GPU_BVH | RT_SCENE_NO_FILTER)))) {
      box_centers.resize((size_t)n);
      for (int i = 0; i < n; i++) box_centers[(size_t)i] = box_center(b.boxes[(size_t)i]);
   }
   std::vector<rt_bvh_node> filter;
   int filter_depth = 0;
   int filter_min = RT_FILTER_MIN_PRIMS;
   if (const char *e = getenv("RT_FILTER_MIN_PRIMS")) { if (atoi(e) >= 2) filter_min = atoi(e); }     /* tests lower it to put small scenes through the filter */
   const bool want_filter = n >= filter_min && !(flags & (RT_SCENE_FAST_BVH | RT_SCENE_GPU_BVH | RT_SCENE_NO_FILTER));
   auto build_filter = [&]() { build_filter_tree(b.boxes, box_centers, n, filter, filter_depth); };
   std::thread filter_thread;
   struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{filter_thread};
   bool filter_inline = false;
   /* (a caller that has work of its own to do before it needs the second tree lets the thread run on past this call) */
   std::unique_ptr<PendingFilter> pending;
   if (want_filter && (flags & RT_FLATTEN_DEFER_FILTER)) {
      try {
         pending.reset(new PendingFilter);
         pending->boxes = b.boxes; pending->centers = box_centers; pending->n = n;
         PendingFilter *pf = pending.get();
         pending->worker = std::thread([pf]() { build_filter_tree(pf->boxes, pf->centers, pf->n, pf->nodes, pf->depth); });
      } catch (...) { pending.reset(); }      /* (no memory or no thread for that: built within this call, then) */
   }
   if (want_filter && !pending) {
      try { filter_thread = std::thread(build_filter); } catch (...) { filter_inline = true; }
   }

   /* (worked out while the second tree is built: only its widening, after the join, needs it) */
   /* How far from the origin of the coordinates a ray may start and still use the second tree.  The two trees' box tests
    * round the differences they take from a ray's origin differently (their boxes differ), by a few units in the last place
    * of the origin's coordinates -- 1e-4 per subtraction for a light a thousand units from a scene of size one.  The second
    * tree's boxes are widened by 2^-20 of this reach, eight times that rounding, and a ray that starts further out than
    * the reach does not use the tree (k_trace).  The reach: twice the largest coordinate among the primitives' boxes (where
    * bounce rays start), the lights (where feelers start) and the camera with its image plane (where primaries start), so
    * that a camera may move about. */
   float origin_reach = 0.0f;
   {
      double far = 1.0;
      for (int i = 0; i < n; i++) {
         const Box3 &bx = b.boxes[(size_t)i];
         const float v[6] = {bx.mn.x, bx.mn.y, bx.mn.z, bx.mx.x, bx.mx.y, bx.mx.z};
         for (float c : v) if (std::fabs((double)c) > far && std::isfinite(c)) far = std::fabs((double)c);
      }
      /* (a light or a camera more than a thousand times further out than the primitives is left outside the reach:
       * widening every box for its sake would cost all the other rays the tree's tightness) */
      const double most = 1024.0 * far;
      for (int i = 0; i < d->n_lights; i++)
         for (int j = 0; j < 3; j++) { const double c = std::fabs((double)d->lights[i].pos[j]); if (c > far && std::isfinite(c)) far = c; }
      if (d->has_camera)
         for (int j = 0; j < 3; j++) {
            const double c = std::fabs((double)d->camera.pos[j]) + std::fabs((double)d->camera.right[j]) + std::fabs((double)d->camera.up[j]);
            if (c > far && std::isfinite(c)) far = c;
         }
      far *= 2.0;
      if (far > most) far = most;
      origin_reach = far < 1e30 ? (float)far : 1e30f;
   }
   float origin_pad = origin_reach * 9.5367431640625e-7f;      /* 2^-20 */
   if (const char *e = getenv("RT_ORIGIN_MARGIN")) origin_pad *= (float)atof(e);      /* (for measurements: what the margin costs) */

   std::vector<rt_bvh_node> flat;
   int32_t root = RT_ROOT_EMPTY;
   int depth = 0;
   if (n > 1 && (flags & RT_SCENE_GPU_BVH)) {
      root = 0;      /* the tree is built on the device by the caller (rt_lbvh_build) from prim_boxes */
   } else if (n > 0 && (flags & RT_SCENE_FAST_BVH)) {
      std::vector<TreeNode> pool;
      pool.reserve((size_t)(2 * n));
      int levels = 0; while ((1 << levels) < n) levels++;
      SahBuilder sb{b.boxes, box_centers, order, pool};
      int root_node = sb.build(0, n, levels + 8, 3);
      if (pool[(size_t)root_node].prim >= 0) root = ~pool[(size_t)root_node].prim;
      else { flatten_tree(pool, root_node, flat, &depth); root = 0; }
   } else if (n > 0) {
      /* the sort rounds of formBVH (bvh.cpp:46-72).  A range is (first, second);
       * the left half is queued as (start, pow2) -- NOT (start, start + pow2) --
       * exactly as bvh.cpp:65 does, so it is empty (and skipped, bvh.cpp:126)
       * whenever start > 0: only prefix ranges and right-hand remainders are ever
       * re-sorted.  Replicated on purpose; the tree shape and leaf order, and with
       * them every near-tie outcome, depend on it. */
      typedef std::pair<int, int> Range;
      std::vector<Range> cur, nxt;
      int axis = 0;
      cur.push_back(Range(0, n));
      while (!cur.empty()) {
         for (size_t k = 0; k < cur.size(); k++) {
            int len = cur[k].second - cur[k].first;
            if (len > 1) merge_sort_range(&order[(size_t)cur[k].first], &tmp[0], len, b.centers, axis);
         }
         while (!cur.empty()) {
            int start = cur.back().first, end = cur.back().second;
            cur.pop_back();
            if (end - start > 2) {
               int p2 = 2;
               while (p2 * 2 < end - start) p2 *= 2;
               nxt.push_back(Range(start, p2));
               nxt.push_back(Range(start + p2, end));
            }
         }
         cur.swap(nxt);
         axis = (axis + 1) % 3;
      }

      /* bottom-up pairing: setupBVHNodes (bvh.cpp:179-192), then one
       * convergeBVHNodes per level (bvh.cpp:88-94, 194-214) */
      std::vector<TreeNode> pool;
      pool.reserve((size_t)(2 * n));
      auto leaf = [&](int prim) { pool.push_back(TreeNode{-1, -1, prim, b.boxes[(size_t)prim]}); return (int)pool.size() - 1; };
      auto join = [&](int l, int r) { pool.push_back(TreeNode{l, r, -1, merged(pool[(size_t)l].bb, pool[(size_t)r].bb)}); return (int)pool.size() - 1; };
      int width = (n - 1) / 2 + 1;
      std::vector<int> level((size_t)width), up((size_t)width);
      for (int i = 0; i < width; i++) {
         if (2 * i + 1 < n) { int l = leaf(order[(size_t)(2 * i)]); int r = leaf(order[(size_t)(2 * i + 1)]); level[(size_t)i] = join(l, r); }
         else level[(size_t)i] = leaf(order[(size_t)(2 * i)]);
      }
      for (int left_n = width; left_n > 1; left_n = (left_n - 1) / 2 + 1) {
         int m = (left_n - 1) / 2 + 1;
         for (int i = 0; i < m; i++) {
            if (2 * i + 1 < left_n) up[(size_t)i] = join(level[(size_t)(2 * i)], level[(size_t)(2 * i + 1)]);
            else up[(size_t)i] = level[(size_t)(2 * i)];
         }
         level.swap(up);
      }
      int root_node = level[0];
      if (pool[(size_t)root_node].prim >= 0) {
         root = ~pool[(size_t)root_node].prim;    /* a one-primitive scene: the root is a leaf */
      } else {
         flatten_tree(pool, root_node, flat, &depth);
         root = 0;
      }
   }

   if (filter_inline) build_filter();
   if (filter_thread.joinable()) filter_thread.join();
   widen_filter_tree(filter, filter_depth, origin_pad);
   if (pending) pending->origin_pad = origin_pad;

   std::vector<rt_dev_light> lights((size_t)d->n_lights);
   for (int i = 0; i < d->n_lights; i++) {
      memset(&lights[(size_t)i], 0, sizeof(rt_dev_light));
      memcpy(lights[(size_t)i].pos, d->lights[i].pos, 12);
      memcpy(lights[(size_t)i].clr, d->lights[i].clr, 12);
   }

   out->n_prims = n; out->n_planes = d->n_planes; out->n_lights = d->n_lights;
   out->n_nodes = (int32_t)flat.size(); out->n_materials = (int32_t)b.mats.materials.size();
   out->root = root; out->tree_depth = depth; out->any_xform = any_xform;
   out->bounds_in_range = 1;
   {
      /* per sphere: the smallest half-axis it has in the world is at least radius * sqrt(3) / |inv_trans| (Frobenius norm
       * of the 3x3 part, which bounds the largest stretch of inv_trans), and |trans| |inv_trans| / 3 says how unevenly the
       * transform stretches (1 for rotations and uniform scalings) */
      double coeff = 0.0;
      for (int i = 0; i < d->n_spheres; i++) {
         const rt_sphere &sp = d->spheres[i];
         double nt = 0.0, ni = 0.0;
         for (int c = 0; c < 3; c++)
            for (int r = 0; r < 3; r++) { nt += (double)sp.mod.trans[c * 4 + r] * sp.mod.trans[c * 4 + r]; ni += (double)sp.mod.inv_trans[c * 4 + r] * sp.mod.inv_trans[c * 4 + r]; }
         nt = std::sqrt(nt); ni = std::sqrt(ni);
         double cond = nt * ni / 3.0; if (!(cond >= 1.0)) cond = 1.0;
         const double rmin = std::fabs((double)sp.r) * std::sqrt(3.0) / ni;
         const double k = cond * cond / rmin;
         if (!(k <= 3.0e38)) coeff = 3.0e38; else if (k > coeff) coeff = k;
      }
      out->sphere_coeff = (float)coeff;
      out->origin_reach = origin_reach;
   }
   for (size_t k = 0; k < flat.size(); k++) {
      const float *v = flat[k].lmin;   /* lmin, lmax, rmin, rmax are 12 consecutive floats */
      for (int j = 0; j < 12; j++)
         if (!(std::fabs(v[j]) <= 1.099511627776e12f)) out->bounds_in_range = 0;
   }
   out->n_filter_nodes = (int32_t)filter.size(); out->filter_depth = filter_depth;
   out->filter_nodes = copy_out(filter);
   out->nodes = copy_out(flat);
   out->prims = b.prims; out->xforms = b.xforms; out->shades = b.shades;      /* handed over as they are */
   b.prims = nullptr; b.xforms = nullptr; b.shades = nullptr;
   out->planes = copy_out(planes);
   out->materials = copy_out(b.mats.materials);
   out->lights = copy_out(lights);
   order.resize((size_t)n);
   out->leaf_order = copy_out(order);
   {
      std::vector<float> pb((size_t)n * 6);
      for (int i = 0; i < n; i++) { put(&pb[(size_t)i * 6], b.boxes[(size_t)i].mn); put(&pb[(size_t)i * 6 + 3], b.boxes[(size_t)i].mx); }
      out->prim_boxes = copy_out(pb);
   }
   {
      std::vector<rt_dev_texinfo> info((size_t)d->n_textures);
      std::vector<uint8_t> texels;
      for (int i = 0; i < d->n_textures; i++) {
         const rt_texture &t = d->textures[i];
         memset(&info[(size_t)i], 0, sizeof(rt_dev_texinfo));
         if (!t.rgba || t.width <= 0 || t.height <= 0) continue;      /* named but never referenced */
         info[(size_t)i].offset = (uint32_t)(texels.size() / 4);
         info[(size_t)i].width = t.width; info[(size_t)i].height = t.height;
         texels.insert(texels.end(), t.rgba, t.rgba + (size_t)t.width * (size_t)t.height * 4);
      }
      out->n_textures = d->n_textures;
      out->tex_info = copy_out(info);
      out->texels = copy_out(texels);
      out->n_texels = texels.size() / 4;
   }
   out->pending_filter = pending.release();
   return RT_OK;
}

/* The second tree of a scene flattened with RT_FLATTEN_DEFER_FILTER: waits for its thread and puts the tree where a
 * call without the flag would have put it.  Does nothing for any other scene, or the second time. */
extern "C" int rt_flat_scene_finish(rt_flat_scene *fs)
{
   if (!fs || !fs->pending_filter) return RT_OK;
   std::unique_ptr<PendingFilter> pf((PendingFilter *)fs->pending_filter);
   fs->pending_filter = nullptr;
   if (pf->worker.joinable()) pf->worker.join();
   widen_filter_tree(pf->nodes, pf->depth, pf->origin_pad);
   free(fs->filter_nodes);
   fs->filter_nodes = copy_out(pf->nodes);
   fs->n_filter_nodes = (int32_t)pf->nodes.size(); fs->filter_depth = pf->depth;
   return RT_OK;
}

extern "C" void rt_flat_scene_free(rt_flat_scene *fs)
{
   if (!fs) return;
   delete (PendingFilter *)fs->pending_filter;      /* (a second tree nobody waited for: its thread is joined here) */
   free(fs->nodes); free(fs->filter_nodes); free(fs->prims); free(fs->xforms); free(fs->shades); free(fs->planes);
   free(fs->materials); free(fs->lights); free(fs->leaf_order); free(fs->tex_info); free(fs->texels); free(fs->prim_boxes);
   memset(fs, 0, sizeof *fs);
}
