// Pose "stickman" rasteriser on the GPU (lib/utils.py:325-512 make_joint_img; data/human36m.py:808-848).
//
// The reference draws each frame on the CPU with cv2.fillPoly / cv2.line, one Python call per
// primitive.  Here every pixel decides for itself: a workgroup first reduces the frame's draw list to
// integer parameters in LDS (clipped, left-to-right-normalised Bresenham lines; the polygon's 16.16
// fixed-point edge table), then each thread evaluates, for four adjacent pixels, the *closed forms* of
// the two OpenCV primitives -- "is (x, y) the pixel LineIterator visits at its major-axis step?" and
// "does (x, y) fall in an even-odd scan-line span?" -- and keeps the colour of the last command that
// covers it (draw order = overwrite order).  Command kinds: 0 body polygon, 1 line between two joints, 4 head line
// (a line that also feeds the "throat length"), 2 neck line (midpoint of the two shoulders -> head joint, drawn
// when the model has no head lines, lib/utils.py:407-433), 3 face line (drawn only if shorter than the throat
// length, :468-505).  Pure integer work (plus the double-precision clip intersection OpenCV itself uses and the
// float64 midpoint / Euclidean lengths of the neck and face rules): bit-exact against the sequential restatement in
// oracle/stickman_oracle.c.  Stores are 4 bytes per lane, 256 B per wavefront; the optional fp32 output
// applies ToTensor and *2-1 (data/base_dataset.py:183-190, data/__init__.py:23-24).
#include "common.h"

#define R_MAXCMD 32
#define R_MAXV 8
#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

struct RLine {
  int x0, y0, dmaj, dmin, ystep, major_is_y, valid;
};

struct RPoly {
  int valid, fill_valid, nv, ne, ymin, ymax;
  int ey0[R_MAXV], ey1[R_MAXV];
  long long ex[R_MAXV], edx[R_MAXV];
  RLine bl[R_MAXV];
};

// OpenCV clipLine (integer region codes, double-precision intersections)
__device__ bool r_clip(long long w, long long h, long long& x1, long long& y1, long long& x2, long long& y2) {
  const long long right = w - 1, bottom = h - 1;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

// LineIterator(pt1, pt2, 8-connected, leftToRight) reduced to its parameters
__device__ RLine r_make_line(long long ax, long long ay, long long bx, long long by, int W, int H) {
  RLine L;
  L.valid = r_clip(W, H, ax, ay, bx, by) ? 1 : 0;
  int dx = (int)(bx - ax), dy = (int)(by - ay);
  int x = (int)ax, y = (int)ay;
  if (dx < 0) { dx = -dx; dy = -dy; x = (int)bx; y = (int)by; }
  L.ystep = dy < 0 ? -1 : 1;
  if (dy < 0) dy = -dy;
  L.major_is_y = dy > dx;
  L.dmaj = L.major_is_y ? dy : dx;
  L.dmin = L.major_is_y ? dx : dy;
  L.x0 = x;
  L.y0 = y;
  return L;
}

// closed form of the iterator: after i major steps the minor coordinate has advanced by
// floor((2*dmin*i + dmaj - 1) / (2*dmaj))   (checked exhaustively against the err recurrence)
__device__ __forceinline__ bool r_on_line(const RLine& L, int x, int y) {
  if (!L.valid) return false;
  const int i = L.major_is_y ? (y - L.y0) * L.ystep : x - L.x0;
  if (i < 0 || i > L.dmaj) return false;
  const int m = L.dmaj ? (2 * L.dmin * i + L.dmaj - 1) / (2 * L.dmaj) : 0;
  return L.major_is_y ? (x == L.x0 + m) : (y == L.y0 + L.ystep * m);
}

__device__ bool r_in_poly(const RPoly& P, int x, int y, int H) {
  if (!P.valid) return false;
  for (int i = 0; i < P.nv; ++i)
    if (r_on_line(P.bl[i], x, y)) return true;  // CollectPolyEdges draws every edge with Line()
  if (!P.fill_valid || y < P.ymin || y >= min(P.ymax, H) || y < 0) return false;
  long long xs[R_MAXV];
  int n = 0;
  for (int e = 0; e < P.ne; ++e)
    if (P.ey0[e] <= y && y < P.ey1[e]) xs[n++] = P.ex[e] + (long long)(y - P.ey0[e]) * P.edx[e];
  for (int a = 1; a < n; ++a) {  // insertion sort, n <= 8
    const long long v = xs[a];
    int b = a - 1;
    while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; --b; }
    xs[b + 1] = v;
  }
  for (int k = 0; k + 1 < n; k += 2) {
    const int x1 = (int)((xs[k] + XY_ONE - 1) >> XY_SHIFT), x2 = (int)(xs[k + 1] >> XY_SHIFT);
    if (x1 <= x && x <= x2) return true;
  }
  return false;
}


// ---- thick lines (cv2.line thickness > 1, lib/utils.py:334-339): OpenCV's ThickLine = FillConvexPoly of a quad in 16.16
// fixed point (outline by Line2, then the two-chain scan line) + a filled Circle at both end points, restated as per-pixel
// closed forms.  One thread per command walks FillConvexPoly's edge chains once and records, per chain, the rows at which a
// new edge segment starts and its (x, dx): a pixel then evaluates x_chain(y) = xs + (y - ybeg) * dx directly.  The circle's
// midpoint iteration only depends on the radius: one half-width table per frame.
#define R_MAXSEG 5
#define R_MAXRAD 40
struct RLine2 {
  int valid, xmajor, m0, ecount, ex, ey;
  long long f0, step;
};
struct RThick {
  int quad, fill, ymin, ylast;
  int nseg[2];
  int ybeg[2][R_MAXSEG];
  long long xs[2][R_MAXSEG], dx[2][R_MAXSEG];
  RLine2 ol[4];
  int cx[2], cy[2];
};

// OpenCV Line2 between two 16.16 points reduced to its parameters
__device__ RLine2 r_make_line2(long long x1, long long y1, long long x2, long long y2, int W, int H) {
  RLine2 L;
  L.valid = r_clip((long long)W << XY_SHIFT, (long long)H << XY_SHIFT, x1, y1, x2, y2) ? 1 : 0;
  long long dx = x2 - x1, dy = y2 - y1;
  const long long j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
  const long long i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
  L.xmajor = ax > ay;
  if (L.xmajor) {
    dy = (dy ^ j) - j;
    if (j) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    L.step = (dy << XY_SHIFT) / (ax | 1);
    L.ecount = (int)((x2 - x1) >> XY_SHIFT);
    L.m0 = (int)((x1 + (XY_ONE >> 1)) >> XY_SHIFT);
    L.f0 = y1 + (XY_ONE >> 1);
  } else {
    dx = (dx ^ i) - i;
    if (i) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    L.step = (dx << XY_SHIFT) / (ay | 1);
    L.ecount = (int)((y2 - y1) >> XY_SHIFT);
    L.m0 = (int)((y1 + (XY_ONE >> 1)) >> XY_SHIFT);
    L.f0 = x1 + (XY_ONE >> 1);
  }
  L.ex = (int)((x2 + (XY_ONE >> 1)) >> XY_SHIFT);
  L.ey = (int)((y2 + (XY_ONE >> 1)) >> XY_SHIFT);
  return L;
}

__device__ __forceinline__ bool r_on_line2(const RLine2& L, int x, int y) {
  if (!L.valid) return false;
  if (x == L.ex && y == L.ey) return true;
  const int k = (L.xmajor ? x : y) - L.m0;
  if (k < 0 || k > L.ecount) return false;
  const int minor = (int)((L.f0 + (long long)k * L.step) >> XY_SHIFT);
  return L.xmajor ? (y == minor) : (x == minor);
}

// ThickLine's quad + FillConvexPoly's chain walk for the segment (ax, ay) -> (bx, by) (integer pixel coordinates)
__device__ void r_make_thick(RThick& T, long long ax, long long ay, long long bx, long long by, int thickness, int W, int H) {
  const long long p0x = ax << XY_SHIFT, p0y = ay << XY_SHIFT, p1x = bx << XY_SHIFT, p1y = by << XY_SHIFT;
  T.cx[0] = (int)((p0x + (XY_ONE >> 1)) >> XY_SHIFT);
  T.cy[0] = (int)((p0y + (XY_ONE >> 1)) >> XY_SHIFT);
  T.cx[1] = (int)((p1x + (XY_ONE >> 1)) >> XY_SHIFT);
  T.cy[1] = (int)((p1y + (XY_ONE >> 1)) >> XY_SHIFT);
  T.quad = T.fill = 0;
  T.nseg[0] = T.nseg[1] = 0;
  const double dx = (double)(p0x - p1x) * (1.0 / XY_ONE), dy = (double)(p1y - p0y) * (1.0 / XY_ONE);
  double r = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
  const int odd = thickness & 1;
  const int th = thickness << (XY_SHIFT - 1);
  if (!(fabs(r) > 2.220446049250313e-16)) return;
  r = __ddiv_rn((double)th + odd * XY_ONE * 0.5, __dsqrt_rn(r));
  const long long dpx = __double2ll_rn(__dmul_rn(dy, r)), dpy = __double2ll_rn(__dmul_rn(dx, r));   // cvRound: half to even
  long long vx[4] = {p0x + dpx, p0x - dpx, p1x - dpx, p1x + dpx};
  long long vy[4] = {p0y + dpy, p0y - dpy, p1y - dpy, p1y + dpy};
  T.quad = 1;
  const int delta = XY_ONE >> 1;
  int imin = 0;
  long long xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
  long long qx = vx[3], qy = vy[3];
  for (int i = 0; i < 4; ++i) {
    if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
    ymax = max(ymax, vy[i]);
    xmax = max(xmax, vx[i]);
    xmin = min(xmin, vx[i]);
    T.ol[i] = r_make_line2(qx, qy, vx[i], vy[i], W, H);
    qx = vx[i];
    qy = vy[i];
  }
  xmin = (xmin + delta) >> XY_SHIFT;
  xmax = (xmax + delta) >> XY_SHIFT;
  ymin = (ymin + delta) >> XY_SHIFT;
  ymax = (ymax + delta) >> XY_SHIFT;
  if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= W || (int)ymin >= H) return;
  if (ymax > H - 1) ymax = H - 1;
  // the two edge chains from the top vertex: chain 0 walks the vertices forwards, chain 1 backwards
  int idxc[2] = {imin, imin}, yec[2], di[2] = {1, 3};
  int y = (int)ymin, edges = 4;
  yec[0] = yec[1] = y;
  T.ymin = y;
  T.ylast = y - 1;
  do {
    for (int i = 0; i < 2; ++i) {
      if (y >= yec[i]) {
        int idx0 = idxc[i], idx = idx0 + di[i];
        if (idx >= 4) idx -= 4;
        for (; edges-- > 0;) {
          const int ty = (int)((vy[idx] + delta) >> XY_SHIFT);
          if (ty > y) {
            const long long xs = vx[idx0], xe = vx[idx];
            yec[i] = ty;
            const int s = T.nseg[i] < R_MAXSEG ? T.nseg[i]++ : R_MAXSEG - 1;
            T.ybeg[i][s] = y;
            T.xs[i][s] = xs;
            T.dx[i][s] = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
            idxc[i] = idx;
            break;
          }
          idx0 = idx;
          idx += di[i];
          if (idx >= 4) idx -= 4;
        }
      }
    }
    if (edges < 0) break;
    T.ylast = y;
  } while (++y <= (int)ymax);
  T.fill = T.nseg[0] > 0 && T.nseg[1] > 0;
}

__device__ bool r_in_thick(const RThick& T, const int* __restrict__ hw, int radius, int x, int y, int W) {
  // end caps: filled circles (rows cy +- ry with |x - cx| <= hw[ry])
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int ry = abs(y - T.cy[e]);
    if (ry <= radius && abs(x - T.cx[e]) <= hw[ry]) return true;
  }
  if (!T.quad) return false;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (r_on_line2(T.ol[i], x, y)) return true;
  if (!T.fill || y < T.ymin || y > T.ylast || y < 0) return false;
  long long xe[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int s = 0;
    for (int q = 1; q < T.nseg[i]; ++q)
      if (T.ybeg[i][q] <= y) s = q;
    xe[i] = T.xs[i][s] + (long long)(y - T.ybeg[i][s]) * T.dx[i][s];
  }
  const long long lo = min(xe[0], xe[1]), hi = max(xe[0], xe[1]);
  const int xx1 = (int)((lo + (XY_ONE >> 1)) >> XY_SHIFT), xx2 = (int)((hi + (XY_ONE >> 1)) >> XY_SHIFT);
  return xx2 >= 0 && xx1 < W && xx1 <= x && x <= xx2;
}

__global__ __launch_bounds__(256) void stickman_raster_kernel(const float* __restrict__ kps, int J,
                                                              const int* __restrict__ body, int n_body,
                                                              const int* __restrict__ cmds, int n_cmds,
                                                              uint8_t* __restrict__ out_u8, float* __restrict__ out_f32,
                                                              int H, int W, int thickness) {
  extern __shared__ __attribute__((aligned(16))) unsigned char thick_raw[];   // RThick[n_cmds] when thickness > 1
  RThick* const thick = reinterpret_cast<RThick*>(thick_raw);
  __shared__ int chw[R_MAXRAD + 1];   // filled-circle half widths per row offset (thickness > 1)
  const int radius = thickness > 1 ? ((thickness << (XY_SHIFT - 1)) + (XY_ONE >> 1)) >> XY_SHIFT : 0;
  __shared__ RLine lines[R_MAXCMD];
  __shared__ RPoly poly;
  __shared__ int ckind[R_MAXCMD], cplane[R_MAXCMD], ccolor[R_MAXCMD];
  __shared__ double clen[R_MAXCMD];   // Euclidean length of the command's segment (float64, as np.linalg.norm)
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* k = kps + (size_t)b * J * 2;

  if (tid < n_cmds) {
    const int* c = cmds + 6 * tid;    // {kind, joint a, joint b, joint c, plane, colour}
    ckind[tid] = c[0];
    cplane[tid] = c[4];
    ccolor[tid] = c[5];
    RLine L;
    L.valid = 0;
    double len = 0.0;
    if (c[0] == 1 || c[0] == 3 || c[0] == 4) {
      const float ax = k[2 * c[1]], ay = k[2 * c[1] + 1], bx = k[2 * c[2]], by = k[2 * c[2] + 1];
      if (ax >= 0.f && ay >= 0.f && bx >= 0.f && by >= 0.f) {
        L = r_make_line((long long)ax, (long long)ay, (long long)bx, (long long)by, W, H);
        if (thickness > 1) {
          r_make_thick(thick[tid], (long long)ax, (long long)ay, (long long)bx, (long long)by, thickness, W, H);
          L.valid |= 1;   // a thick line is drawn whether or not its centre line survives the clip
        }
        const double dx = (double)ax - (double)bx, dy = (double)ay - (double)by;
        len = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
        if (c[0] == 3) L.valid |= 2;   // face line: still waits for the throat-length test (bit 1)
        if (c[0] == 4) L.valid |= 4;   // head line: clipped away or not, a valid one feeds the throat length (bit 2)
      }
    } else if (c[0] == 2) {
      // neck = 0.5 * (rshoulder + lshoulder) if neither has a negative coordinate, else (-1, -1)  (:408-413)
      const float rx = k[2 * c[1]], ry = k[2 * c[1] + 1], lx = k[2 * c[2]], ly = k[2 * c[2] + 1];
      const float hx = k[2 * c[3]], hy = k[2 * c[3] + 1];
      if (rx >= 0.f && ry >= 0.f && lx >= 0.f && ly >= 0.f && hx >= 0.f && hy >= 0.f) {
        const double nx = 0.5 * ((double)rx + (double)lx), ny = 0.5 * ((double)ry + (double)ly);
        L = r_make_line((long long)nx, (long long)ny, (long long)hx, (long long)hy, W, H);
        if (thickness > 1) {
          r_make_thick(thick[tid], (long long)nx, (long long)ny, (long long)hx, (long long)hy, thickness, W, H);
          L.valid |= 1;
        }
        const double dx = nx - (double)hx, dy = ny - (double)hy;
        len = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
        L.valid |= 4;                  // drawn or not, a valid neck segment defines the throat length
      }
    }
    lines[tid] = L;
    clen[tid] = len;
  }
  if (tid == 254 && thickness > 1) {   // OpenCV Circle(fill): the midpoint iteration, widest span per row offset
    for (int r = 0; r <= radius; ++r) chw[r] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
      chw[dy] = max(chw[dy], dx);
      chw[dx] = max(chw[dx], dy);
      dy++;
      err += plus;
      plus += 2;
      const int mask = (err <= 0) - 1;
      err -= minus & mask;
      dx += mask;
      minus -= mask & 2;
    }
  }
  if (tid == 255) {  // the body polygon (shared by every polygon command)
    RPoly P;
    P.valid = 0;
    P.fill_valid = 0;
    P.nv = P.ne = 0;
    long long vx[R_MAXV], vy[R_MAXV];
    int n = 0;
    if (n_body > 2)
      for (int i = 0; i < n_body && n < R_MAXV; ++i) {
        const float x = k[2 * body[i]], y = k[2 * body[i] + 1];
        if (x >= 0.f && y >= 0.f) { vx[n] = (long long)x; vy[n] = (long long)y; ++n; }
      }
    if (n > 2) {
      P.valid = 1;
      P.nv = n;
      long long px = vx[n - 1] << XY_SHIFT, py = vy[n - 1];
      int ymax = INT_MIN, ymin = INT_MAX;
      long long xmax = -1, xmin = LLONG_MAX;
      for (int i = 0; i < n; ++i) {
        const long long qx = vx[i] << XY_SHIFT, qy = vy[i];
        P.bl[i] = r_make_line((px + (XY_ONE >> 1)) >> XY_SHIFT, py, (qx + (XY_ONE >> 1)) >> XY_SHIFT, qy, W, H);
        if (py != qy) {
          const int e = P.ne++;
          if (py < qy) { P.ey0[e] = (int)py; P.ey1[e] = (int)qy; P.ex[e] = px; }
          else { P.ey0[e] = (int)qy; P.ey1[e] = (int)py; P.ex[e] = qx; }
          P.edx[e] = (qx - px) / (qy - py);
          const long long x1 = P.ex[e] + (long long)(P.ey1[e] - P.ey0[e]) * P.edx[e];
          ymin = min(ymin, P.ey0[e]);
          ymax = max(ymax, P.ey1[e]);
          xmin = min(xmin, min(P.ex[e], x1));
          xmax = max(xmax, max(P.ex[e], x1));
        }
        px = qx;
        py = qy;
      }
      P.ymin = ymin;
      P.ymax = ymax;
      P.fill_valid = P.ne >= 2 && !(ymax < 0 || ymin >= H || xmax < 0 || xmin >= ((long long)W << XY_SHIFT));
    }
    poly = P;
  }
  __syncthreads();
  if (tid == 0) {   // throat length = longest valid head / neck segment (:415-417, :466-467); then the face-line test
    double throat = 0.0;
    for (int c = 0; c < n_cmds; ++c)
      if ((ckind[c] == 2 || ckind[c] == 4) && (lines[c].valid & 4)) throat = fmax(throat, clen[c]);
    for (int c = 0; c < n_cmds; ++c) {
      int v = lines[c].valid;
      if (ckind[c] == 3) v = ((v & 2) && clen[c] < throat) ? (v & 1) : 0;
      lines[c].valid = v & 1;
    }
  }
  __syncthreads();

  const int W4 = W >> 2;  // W % 4 == 0 (checked by the host)
  for (int q = blockIdx.y * 256 + tid; q < H * W4; q += gridDim.y * 256) {
    const int y = q / W4, x0 = (q - y * W4) * 4;
    uint32_t packed[3] = {0u, 0u, 0u};
#pragma unroll
    for (int dxp = 0; dxp < 4; ++dxp) {
      const int x = x0 + dxp;
      int val[3] = {0, 0, 0};
      const bool inp = r_in_poly(poly, x, y, H);
      for (int c = 0; c < n_cmds; ++c) {
        const bool hit = ckind[c] == 0 ? inp
                         : (thickness > 1 ? (lines[c].valid && r_in_thick(thick[c], chw, radius, x, y, W))
                                          : r_on_line(lines[c], x, y));
        if (hit) val[cplane[c]] = ccolor[c];
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) packed[p] |= (uint32_t)(val[p] & 255) << (8 * dxp);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const size_t o = (((size_t)b * 3 + p) * H + y) * W + x0;
      if (out_u8) *reinterpret_cast<uint32_t*>(out_u8 + o) = packed[p];
      if (out_f32) {
        float4 f;
        f.x = ((float)(packed[p] & 255u) / 255.0f) * 2.0f - 1.0f;
        f.y = ((float)((packed[p] >> 8) & 255u) / 255.0f) * 2.0f - 1.0f;
        f.z = ((float)((packed[p] >> 16) & 255u) / 255.0f) * 2.0f - 1.0f;
        f.w = ((float)(packed[p] >> 24) / 255.0f) * 2.0f - 1.0f;
        *reinterpret_cast<float4*>(out_f32 + o) = f;
      }
    }
  }
}

extern "C" int vunet_stickman_raster_thick(const float* kps, int32_t B, int32_t J, const int32_t* body, int32_t n_body,
                                           const int32_t* cmds, int32_t n_cmds, uint8_t* out_u8, float* out_f32, int32_t H,
                                           int32_t W, int32_t thickness, void* stream) {
  if (!kps || !cmds || (!out_u8 && !out_f32) || B < 1 || J < 1 || H < 1 || W < 4 || (W & 3)) return VUNET_ERR_ARG;
  if (n_cmds < 0 || n_cmds > R_MAXCMD || n_body < 0 || n_body > R_MAXV || (n_body > 0 && !body)) return VUNET_ERR_ARG;
  if (thickness < 1 || thickness > 2 * R_MAXRAD - 1) return VUNET_ERR_ARG;
  int gy = (H * (W >> 2) + 255) / 256;
  if (gy > 64) gy = 64;
  const size_t lds = thickness > 1 ? (size_t)n_cmds * sizeof(RThick) : 0;
  VUNET_LAUNCH(stickman_raster_kernel, dim3((unsigned)B, (unsigned)gy), dim3(256), lds, (hipStream_t)stream, kps, J, body,
               n_body, cmds, n_cmds, out_u8, out_f32, H, W, thickness);
  return vunet_check_launch();
}

extern "C" int vunet_stickman_raster(const float* kps, int32_t B, int32_t J, const int32_t* body, int32_t n_body,
                                     const int32_t* cmds, int32_t n_cmds, uint8_t* out_u8, float* out_f32, int32_t H,
                                     int32_t W, void* stream) {
  return vunet_stickman_raster_thick(kps, B, J, body, n_body, cmds, n_cmds, out_u8, out_f32, H, W, 1, stream);
}
