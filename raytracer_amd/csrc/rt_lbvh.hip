/*
 * rt_lbvh.hip -- opt-in device-side BVH construction (RT_SCENE_GPU_BVH).
 *
 * NOT the reference's tree (formBVH, bvh.cpp:30-223, stays the default and the only parity mode): a linear BVH
 * over the primitives' bounding boxes -- Morton codes of the box centres, one radix sort, the radix tree of
 * Karras ("Maximizing parallelism in the construction of BVHs, octrees and k-d trees", 2012), boxes fitted
 * bottom-up -- written straight into the 64-byte node records the traversal kernel reads (both children's
 * boxes in the parent, child < 0 = leaf ~primitive).  Build time is milliseconds for a million primitives,
 * against seconds for either host builder; tree quality is below the host SAH tree's.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "rt_common.h"

#define LB_BLOCK 256

namespace {

struct Box6 { float mn[3], mx[3]; };

__device__ __forceinline__ unsigned expand10(unsigned v)
{
   v &= 0x3ffu;
   v = (v | (v << 16)) & 0x030000ffu;
   v = (v | (v << 8)) & 0x0300f00fu;
   v = (v | (v << 4)) & 0x030c30c3u;
   v = (v | (v << 2)) & 0x09249249u;
   return v;
}

/* key = 30-bit Morton code of the box centre, made unique by the primitive index in the low word */
__global__ __launch_bounds__(LB_BLOCK) void k_lbvh_keys(const Box6 *boxes, int n, float3 cmin, float3 cinv, unsigned long long *keys)
{
   int i = blockIdx.x * LB_BLOCK + threadIdx.x;
   if (i >= n) return;
   Box6 b = boxes[i];
   float x = (0.5f * (b.mn[0] + b.mx[0]) - cmin.x) * cinv.x;
   float y = (0.5f * (b.mn[1] + b.mx[1]) - cmin.y) * cinv.y;
   float z = (0.5f * (b.mn[2] + b.mx[2]) - cmin.z) * cinv.z;
   unsigned xi = (unsigned)fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
   unsigned yi = (unsigned)fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
   unsigned zi = (unsigned)fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
   unsigned code = (expand10(xi) << 2) | (expand10(yi) << 1) | expand10(zi);
   keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

/* length of the common prefix of keys i and j, -1 outside the array */
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j)
{
   if (j < 0 || j >= n) return -1;
   return __clzll((long long)(keys[i] ^ keys[j]));
}

/* one internal node per thread: its key range and split (Karras 2012, section 4).  Children: >= 0 internal node,
 * < 0 leaf at sorted position ~child. */
__global__ __launch_bounds__(LB_BLOCK) void k_lbvh_tree(const unsigned long long *keys, int n, int2 *children, int *parent_of_node,
                                                       int *parent_of_leaf)
{
   int i = blockIdx.x * LB_BLOCK + threadIdx.x;
   if (i >= n - 1) return;
   int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
   int dmin = delta(keys, n, i, i - d);
   int lmax = 2;
   while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
   int l = 0;
   for (int t = lmax / 2; t >= 1; t /= 2)
      if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
   int j = i + l * d;
   int dnode = delta(keys, n, i, j);
   int s = 0;
   for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
      if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
      if (t <= 1) break;
   }
   int gamma = i + s * d + (d < 0 ? d : 0);
   int lo = i < j ? i : j, hi = i < j ? j : i;
   int2 ch;
   if (lo == gamma) { ch.x = ~gamma; parent_of_leaf[gamma] = i; } else { ch.x = gamma; parent_of_node[gamma] = i; }
   if (hi == gamma + 1) { ch.y = ~(gamma + 1); parent_of_leaf[gamma + 1] = i; } else { ch.y = gamma + 1; parent_of_node[gamma + 1] = i; }
   children[i] = ch;
   if (i == 0) parent_of_node[0] = -1;
}

__device__ __forceinline__ Box6 box_union(const float *amn, const float *amx, const float *bmn, const float *bmx)
{
   Box6 r;
   for (int k = 0; k < 3; k++) { r.mn[k] = fminf(amn[k], bmn[k]); r.mx[k] = fmaxf(amx[k], bmx[k]); }
   return r;
}

/* one thread per leaf walks towards the root; the second thread to reach a node finds both subtrees finished and
 * writes the node's record: its children's boxes and references (leaf = ~primitive id) */
__global__ __launch_bounds__(LB_BLOCK) void k_lbvh_fit(const unsigned long long *keys, const Box6 *boxes, int n, const int2 *children,
                                                      const int *parent_of_node, const int *parent_of_leaf, unsigned *arrived,
                                                      int *depth_of_node, rt_bvh_node *nodes)
{
   int k = blockIdx.x * LB_BLOCK + threadIdx.x;
   if (k >= n) return;
   int p = parent_of_leaf[k];
   while (p >= 0) {
      __threadfence();
      if (atomicAdd(&arrived[p], 1u) == 0u) return;           /* the sibling subtree is not done yet */
      __threadfence();
      int2 ch = children[p];
      rt_bvh_node rec;
      int depth = 0;
      for (int side = 0; side < 2; side++) {
         int c = side ? ch.y : ch.x;
         Box6 b;
         int ref;
         if (c < 0) {
            int prim = (int)(unsigned)(keys[~c] & 0xffffffffULL);
            b = boxes[prim];
            ref = ~prim;
         } else {
            const volatile rt_bvh_node *cn = nodes + c;
            float lmn[3] = {cn->lmin[0], cn->lmin[1], cn->lmin[2]}, lmx[3] = {cn->lmax[0], cn->lmax[1], cn->lmax[2]};
            float rmn[3] = {cn->rmin[0], cn->rmin[1], cn->rmin[2]}, rmx[3] = {cn->rmax[0], cn->rmax[1], cn->rmax[2]};
            b = box_union(lmn, lmx, rmn, rmx);
            ref = c;
            int dc = ((const volatile int *)depth_of_node)[c];
            depth = dc > depth ? dc : depth;
         }
         float *mn = side ? rec.rmin : rec.lmin, *mx = side ? rec.rmax : rec.lmax;
         for (int q = 0; q < 3; q++) { mn[q] = b.mn[q]; mx[q] = b.mx[q]; }
         if (side) rec.right = ref; else rec.left = ref;
      }
      rec.pad[0] = rec.pad[1] = 0;
      nodes[p] = rec;
      depth_of_node[p] = depth + 1;
      p = parent_of_node[p];
   }
}

} // namespace

#define LB_TRY(expr)                                                                              \
   do {                                                                                           \
      hipError_t e_ = (expr);                                                                     \
      if (e_ != hipSuccess) { rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); rc = RT_ERR_HIP; goto done; } \
   } while (0)

/* boxes_host: n boxes of 6 floats (min, max).  On success *nodes_out (n - 1 records, malloc'ed), *order_out
 * (the primitives in leaf order, malloc'ed) and *depth_out are set.  n >= 2. */
extern "C" int rt_lbvh_build(const float *boxes_host, int n, rt_bvh_node **nodes_out, int32_t **order_out, int *depth_out,
                             float *ms_out)
{
   int rc = RT_OK;
   Box6 *d_boxes = nullptr;
   unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
   int2 *d_children = nullptr;
   int *d_pnode = nullptr, *d_pleaf = nullptr, *d_depth = nullptr;
   unsigned *d_arrived = nullptr;
   rt_bvh_node *d_nodes = nullptr;
   void *d_temp = nullptr;
   size_t temp_bytes = 0;
   hipEvent_t e0 = nullptr, e1 = nullptr;
   rt_bvh_node *nodes = nullptr;
   int32_t *order = nullptr;
   unsigned long long *h_sorted = nullptr;
   unsigned grid = (unsigned)((n + LB_BLOCK - 1) / LB_BLOCK);
   float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
   for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
         float c = 0.5f * (boxes_host[6 * i + k] + boxes_host[6 * i + 3 + k]);
         if (c < cmn[k]) cmn[k] = c;
         if (c > cmx[k]) cmx[k] = c;
      }
   float3 cmin = make_float3(cmn[0], cmn[1], cmn[2]);
   float3 cinv = make_float3(cmx[0] > cmn[0] ? 1.0f / (cmx[0] - cmn[0]) : 0.0f, cmx[1] > cmn[1] ? 1.0f / (cmx[1] - cmn[1]) : 0.0f,
                             cmx[2] > cmn[2] ? 1.0f / (cmx[2] - cmn[2]) : 0.0f);

   LB_TRY(hipMalloc((void **)&d_boxes, sizeof(Box6) * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_keys, 8 * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_sorted, 8 * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_children, sizeof(int2) * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_pnode, 4 * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_pleaf, 4 * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_depth, 4 * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_arrived, 4 * (size_t)n));
   LB_TRY(hipMalloc((void **)&d_nodes, sizeof(rt_bvh_node) * (size_t)n));
   LB_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_sorted, n, 0, 64, (hipStream_t)0));
   LB_TRY(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16));
   LB_TRY(hipMemcpy(d_boxes, boxes_host, sizeof(Box6) * (size_t)n, hipMemcpyHostToDevice));
   LB_TRY(hipEventCreate(&e0)); LB_TRY(hipEventCreate(&e1));
   LB_TRY(hipEventRecord(e0, 0));
   LB_TRY(hipMemsetAsync(d_arrived, 0, 4 * (size_t)n, 0));
   LB_TRY(hipMemsetAsync(d_depth, 0, 4 * (size_t)n, 0));
   hipLaunchKernelGGL(k_lbvh_keys, dim3(grid), dim3(LB_BLOCK), 0, 0, (const Box6 *)d_boxes, n, cmin, cinv, d_keys);
   LB_TRY(hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_sorted, n, 0, 64, (hipStream_t)0));
   hipLaunchKernelGGL(k_lbvh_tree, dim3(grid), dim3(LB_BLOCK), 0, 0, (const unsigned long long *)d_sorted, n, d_children, d_pnode, d_pleaf);
   hipLaunchKernelGGL(k_lbvh_fit, dim3(grid), dim3(LB_BLOCK), 0, 0, (const unsigned long long *)d_sorted, (const Box6 *)d_boxes, n,
                      (const int2 *)d_children, (const int *)d_pnode, (const int *)d_pleaf, d_arrived, d_depth, d_nodes);
   LB_TRY(hipEventRecord(e1, 0));
   LB_TRY(hipGetLastError());
   LB_TRY(hipDeviceSynchronize());
   if (ms_out) (void)hipEventElapsedTime(ms_out, e0, e1);

   nodes = (rt_bvh_node *)malloc(sizeof(rt_bvh_node) * (size_t)(n - 1));
   order = (int32_t *)malloc(4 * (size_t)n);
   h_sorted = (unsigned long long *)malloc(8 * (size_t)n);
   LB_TRY(hipMemcpy(nodes, d_nodes, sizeof(rt_bvh_node) * (size_t)(n - 1), hipMemcpyDeviceToHost));
   LB_TRY(hipMemcpy(h_sorted, d_sorted, 8 * (size_t)n, hipMemcpyDeviceToHost));
   LB_TRY(hipMemcpy(depth_out, d_depth, 4, hipMemcpyDeviceToHost));          /* the root is node 0 */
   for (int i = 0; i < n; i++) order[i] = (int32_t)(unsigned)(h_sorted[i] & 0xffffffffULL);
   *nodes_out = nodes; *order_out = order;
   nodes = nullptr; order = nullptr;
done:
   free(nodes); free(order); free(h_sorted);
   if (e0) (void)hipEventDestroy(e0);
   if (e1) (void)hipEventDestroy(e1);
   (void)hipFree(d_boxes); (void)hipFree(d_keys); (void)hipFree(d_sorted); (void)hipFree(d_children); (void)hipFree(d_pnode);
   (void)hipFree(d_pleaf); (void)hipFree(d_depth); (void)hipFree(d_arrived); (void)hipFree(d_nodes); (void)hipFree(d_temp);
   return rc;
}
