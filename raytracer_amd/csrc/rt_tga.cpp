/*
 * rt_tga.cpp -- frame -> TGA, byte for byte what the reference's driver writes.
 *
 * The reference converts each uchar4 channel k to a double k/255.0, stores it
 * in an Image whose running maximum starts at 1.0, and WriteTga() then emits
 * (unsigned char)((value / max) * 255) (raytrace.cpp:44-55, Image.cpp:10-14,
 * 77-101, 189-205).  max can never exceed 1.0 for byte input; the double round
 * trip (k/255.0)*255 is evaluated as the reference evaluates it (it happens
 * to give back k for all 256 values).
 * Header: 18 bytes, type 2, 24 bits, origin bottom-left; pixels BGR, rows
 * bottom to top -- the frame's row 0 is already the bottom row.
 */
#include "rt_common.h"

#include <cstdio>
#include <vector>

extern "C" int rt_write_tga(const char *path, int width, int height, const uint8_t *rgba)
{
   if (!path || !rgba || width <= 0 || height <= 0 || width > 0xffff || height > 0xffff) {
      rt_set_error("rt_write_tga: bad argument");
      return RT_ERR_BAD_ARGUMENT;
   }
   FILE *fp = fopen(path, "wb");
   if (!fp) { rt_set_error("rt_write_tga: cannot open %s", path); return RT_ERR_IO; }
   unsigned char header[18] = {0};
   header[2] = 2;
   header[12] = (unsigned char)(width & 0xff); header[13] = (unsigned char)((width & 0xff00) >> 8);
   header[14] = (unsigned char)(height & 0xff); header[15] = (unsigned char)((height & 0xff00) >> 8);
   header[16] = 24;
   unsigned char lut[256];
   const double max = 1.0;
   for (int k = 0; k < 256; k++) lut[k] = (unsigned char)(((k / 255.0) / max) * 255);
   std::vector<unsigned char> row((size_t)width * 3);
   bool ok = fwrite(header, 1, 18, fp) == 18;
   for (int y = 0; y < height && ok; y++) {
      const uint8_t *src = rgba + (size_t)y * width * 4;
      for (int x = 0; x < width; x++) {
         row[(size_t)x * 3 + 0] = lut[src[x * 4 + 2]];
         row[(size_t)x * 3 + 1] = lut[src[x * 4 + 1]];
         row[(size_t)x * 3 + 2] = lut[src[x * 4 + 0]];
      }
      ok = fwrite(row.data(), 1, row.size(), fp) == row.size();
   }
   if (fclose(fp) != 0) ok = false;
   if (!ok) { rt_set_error("rt_write_tga: write to %s failed", path); return RT_ERR_IO; }
   return RT_OK;
}
