/* Oracle (TEST INFRASTRUCTURE ONLY — never linked into the product).
 *
 * Plain-C restatement of skimage.measure.label(seg) as called by
 * cellulus/utils/misc.py:16,25 (connectivity=None -> full: 8 in 2-D, 26 in
 * 3-D; background 0; maximal regions of EQUAL value; ids assigned in raster
 * order of each region's first pixel) and of size_filter (misc.py:11-25).
 * Flood fill in raster order — the definition itself, no union-find tricks.
 * scikit-image is unpinned in the reference (pyproject.toml:26); pinned here
 * against scikit-image 0.18.3 golden vectors (tests/golden).
 */
#include <stdlib.h>

/* returns the number of components; out must hold Z*Y*X ints */
int cc_oracle_label(const int* seg, int* out, int Z, int Y, int X) {
  const long n = (long)Z * Y * X;
  for (long i = 0; i < n; ++i) out[i] = 0;
  long* stack = (long*)malloc(sizeof(long) * (size_t)n);
  int next = 0;
  for (long i = 0; i < n; ++i) {
    if (seg[i] == 0 || out[i] != 0) continue;
    const int v = seg[i];
    ++next;
    long top = 0;
    stack[top++] = i;
    out[i] = next;
    while (top > 0) {
      const long p = stack[--top];
      const int x = (int)(p % X), y = (int)((p / X) % Y), z = (int)(p / ((long)X * Y));
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int zz = z + dz, yy = y + dy, xx = x + dx;
            if (zz < 0 || zz >= Z || yy < 0 || yy >= Y || xx < 0 || xx >= X) continue;
            const long q = ((long)zz * Y + yy) * X + xx;
            if (seg[q] == v && out[q] == 0) {
              out[q] = next;
              stack[top++] = q;
            }
          }
    }
  }
  free(stack);
  return next;
}

/* size_filter: label, zero components smaller than min_size IN seg, relabel */
int cc_oracle_size_filter(int* seg, int* out, int Z, int Y, int X, int min_size) {
  const long n = (long)Z * Y * X;
  const int ncomp = cc_oracle_label(seg, out, Z, Y, X);
  long* sizes = (long*)calloc((size_t)ncomp + 1, sizeof(long));
  for (long i = 0; i < n; ++i) sizes[out[i]]++;
  for (long i = 0; i < n; ++i)
    if (out[i] != 0 && sizes[out[i]] < min_size) seg[i] = 0;
  free(sizes);
  return cc_oracle_label(seg, out, Z, Y, X);
}
