/* The sort rounds of the reference's tree build have two implementations in rt_scene_build.cpp: the literal bottom-up
 * merge of bvh.cpp:106-171 (right run first on equal keys) and a radix sort that must give the same permutation.  This
 * program runs both on the same inputs: random keys, keys with many ties, signed zeros, infinities, denormals, ranges of
 * every length around the switch-over, already sorted and reversed inputs, permuted id arrays.  Prints "bad=0" when
 * every permutation agreed. */
#include "../../raytracer_amd/csrc/rt_scene_build.cpp"

#include <cstdint>
#include <random>

int main()
{
   std::mt19937 rng(12345);
   long long bad = 0, tested = 0, with_ties = 0;
   const float specials[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, 1e-45f, -1e-45f, 1e-40f, 3.4e38f, -3.4e38f, 1.17549435e-38f};
   for (int trial = 0; trial < 6000; trial++) {
      int n = trial < 600 ? 150 + trial : (int)(rng() % 20000) + 2;
      int kind = (int)(rng() % 6);
      std::vector<F3> centers((size_t)n);
      for (int i = 0; i < n; i++) {
         float v[3];
         for (int a = 0; a < 3; a++) {
            switch (kind) {
            case 0: v[a] = (float)((double)rng() / 4294967296.0 * 20.0 - 10.0); break;
            case 1: v[a] = (float)(int)(rng() % 7) - 3.0f; break;                                   /* heavy ties */
            case 2: v[a] = specials[rng() % (sizeof specials / sizeof specials[0])]; break;       /* signed zeros, infinities, denormals */
            case 3: v[a] = (float)i * 0.25f; break;                                                 /* sorted */
            case 4: v[a] = (float)(n - i) * 0.25f; break;                                           /* reversed */
            default: { uint32_t u = rng(); if ((u & 0x7f800000u) == 0x7f800000u) u &= 0xff7fffffu; memcpy(&v[a], &u, 4); } break;   /* any finite bit pattern */
            }
         }
         centers[(size_t)i] = F3{v[0], v[1], v[2]};
      }
      std::vector<int32_t> ids((size_t)n), tmp((size_t)n);
      for (int i = 0; i < n; i++) ids[(size_t)i] = i;
      if (rng() & 1) std::shuffle(ids.begin(), ids.end(), rng);
      for (int axis = 0; axis < 3; axis++) {
         std::vector<int32_t> a(ids), b(ids);
         merge_sort_range(a.data(), tmp.data(), n, centers, axis);
         merge_sort_range_literal(b.data(), tmp.data(), n, centers, axis);
         tested++;
         if (a != b) bad++;
         for (int i = 1; i < n; i++)
            if (axis_of(centers[(size_t)b[(size_t)i]], axis) == axis_of(centers[(size_t)b[(size_t)i - 1]], axis)) { with_ties++; break; }
      }
   }
   /* a NaN key sends the range to the merge itself: same function, same answer */
   {
      int n = 1000;
      std::vector<F3> centers((size_t)n);
      for (int i = 0; i < n; i++) centers[(size_t)i] = F3{(float)(rng() % 100), (i % 97 == 0) ? NAN : (float)(rng() % 50), 0.0f};
      std::vector<int32_t> ids((size_t)n), tmp((size_t)n);
      for (int i = 0; i < n; i++) ids[(size_t)i] = i;
      std::vector<int32_t> a(ids), b(ids);
      merge_sort_range(a.data(), tmp.data(), n, centers, 1);
      merge_sort_range_literal(b.data(), tmp.data(), n, centers, 1);
      tested++;
      if (a != b) bad++;
   }
   printf("tested=%lld with_ties=%lld bad=%lld\n", tested, with_ties, bad);
   return bad ? 1 : 0;
}
