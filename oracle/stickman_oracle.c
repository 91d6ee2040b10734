/*
 * stickman_oracle.c -- CPU restatement of the pose "stickman" rasteriser.  TEST INFRASTRUCTURE ONLY
 * (tests/, __graft_entry__.smoke(), bench.py cpu_baseline); the product never links or calls it.
 *
 * Follows lib/utils.py:325-512 (make_joint_img) of the reference at thickness 1, LINE_8, shift 0 (the shipped configs
 * never set stickman_scale) -- every branch of it: per-line colours and the single-channel mode are draw-list
 * parameters, the neck line of the joint models without head lines and the throat-length-gated face lines are command
 * kinds of their own (see the draw list below).  Default colours of the Human3.6m model:
 * body polygon -> cv2.fillPoly on planes (0,1,2) with colours (0,127,255) (:345-355); right limbs ->
 * cv2.line colour 255 on plane 1 (:357-380); left limbs -> 255 on plane 0 (:382-405); head lines -> 127
 * on planes 0 and 1 (:434-462); joints are valid iff both coordinates >= 0 and are truncated with
 * np.int_ (:358-361).
 *
 * cv2 itself is a third-party dependency that is absent here (pinned opencv=4.1.2 in environment.yml,
 * not vendored): PARITY UNPINNED against OpenCV.  The two drawing primitives are restated from the
 * published OpenCV 4.1.2 algorithm (modules/imgproc/src/drawing.cpp): clipLine (integer Cohen-Sutherland
 * with double-precision intersection), LineIterator (8-connected Bresenham, left-to-right), and
 * CollectPolyEdges + FillEdgeCollection (boundary lines, then an active-edge scan line in 16.16 fixed
 * point with incremental x += dx and ceil/floor span ends).  This file keeps their *sequential* structure
 * on purpose: the GPU kernel (csrc/raster.hip) uses closed forms per pixel, so agreement between the two
 * is a real check.  Known-answer tests: tests/test_stickman_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

typedef struct { int64_t x, y; } pt64;

/* OpenCV clipLine(Size2l, Point2l&, Point2l&) */
static int clip_line(int64_t w, int64_t h, pt64* p1, pt64* p2) {
  int c1, c2;
  const int64_t right = w - 1, bottom = h - 1;
  if (w <= 0 || h <= 0) return 0;
  int64_t x1 = p1->x, y1 = p1->y, x2 = p2->x, y2 = p2->y;
  c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int64_t a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (int64_t)((double)(a - y1) * (x2 - x1) / (y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (int64_t)((double)(a - y2) * (x2 - x1) / (y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (int64_t)((double)(a - x1) * (y2 - y1) / (x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (int64_t)((double)(a - x2) * (y2 - y1) / (x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
    p1->x = x1; p1->y = y1; p2->x = x2; p2->y = y2;
  }
  return (c1 | c2) == 0;
}

/* OpenCV Line(): LineIterator(img, pt1, pt2, 8, leftToRight=true), every visited pixel set to colour */
static void draw_line(uint8_t* img, int w, int h, int64_t ax, int64_t ay, int64_t bx, int64_t by, uint8_t color) {
  pt64 p1 = {ax, ay}, p2 = {bx, by};
  if (!clip_line(w, h, &p1, &p2)) return;
  int dx = (int)(p2.x - p1.x), dy = (int)(p2.y - p1.y);
  int x = (int)p1.x, y = (int)p1.y;
  if (dx < 0) { /* start from the left end point */
    dx = -dx;
    dy = -dy;
    x = (int)p2.x;
    y = (int)p2.y;
  }
  int ystep = dy < 0 ? -1 : 1;
  if (dy < 0) dy = -dy;
  int major_is_y = dy > dx;
  int dmaj = major_is_y ? dy : dx, dmin = major_is_y ? dx : dy;
  int err = dmaj - (dmin + dmin);
  const int plus = dmaj + dmaj, minus = -(dmin + dmin);
  const int count = dmaj + 1;
  for (int i = 0; i < count; ++i) {
    img[(size_t)y * w + x] = color;
    const int mask = err < 0;
    err += minus + (mask ? plus : 0);
    if (major_is_y) { y += ystep; if (mask) x += 1; }
    else { x += 1; if (mask) y += ystep; }
  }
}

typedef struct edge { int y0, y1; int64_t x, dx; struct edge* next; } edge;

static int cmp_edges(const void* a, const void* b) {
  const edge *e1 = (const edge*)a, *e2 = (const edge*)b;
  if (e1->y0 != e2->y0) return e1->y0 < e2->y0 ? -1 : 1;
  if (e1->x != e2->x) return e1->x < e2->x ? -1 : 1;
  if (e1->dx != e2->dx) return e1->dx < e2->dx ? -1 : 1;
  return 0;
}

/* OpenCV fillPoly for one contour: CollectPolyEdges (boundary lines + edge table) then FillEdgeCollection */
static void fill_poly(uint8_t* img, int w, int h, const int64_t* vx, const int64_t* vy, int n, uint8_t color) {
  edge edges[16 + 1];
  int total = 0;
  if (n < 1 || n > 16) return;
  int64_t px = vx[n - 1] << XY_SHIFT, py = vy[n - 1];
  for (int i = 0; i < n; ++i) {
    const int64_t qx = vx[i] << XY_SHIFT, qy = vy[i];
    draw_line(img, w, h, (px + (XY_ONE >> 1)) >> XY_SHIFT, py, (qx + (XY_ONE >> 1)) >> XY_SHIFT, qy, color);
    if (py != qy) {
      edge e;
      if (py < qy) { e.y0 = (int)py; e.y1 = (int)qy; e.x = px; }
      else { e.y0 = (int)qy; e.y1 = (int)py; e.x = qx; }
      e.dx = (qx - px) / (qy - py);
      e.next = 0;
      edges[total++] = e;
    }
    px = qx; py = qy;
  }
  if (total < 2) return;
  int y_max = INT32_MIN, y_min = INT32_MAX;
  int64_t x_max = -1, x_min = INT64_MAX;
  for (int i = 0; i < total; ++i) {
    const edge* e = &edges[i];
    const int64_t x1 = e->x + (int64_t)(e->y1 - e->y0) * e->dx;
    if (e->y0 < y_min) y_min = e->y0;
    if (e->y1 > y_max) y_max = e->y1;
    if (e->x < x_min) x_min = e->x;
    if (e->x > x_max) x_max = e->x;
    if (x1 < x_min) x_min = x1;
    if (x1 > x_max) x_max = x1;
  }
  if (y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((int64_t)w << XY_SHIFT)) return;
  qsort(edges, total, sizeof(edge), cmp_edges);
  edge tmp;
  memset(&tmp, 0, sizeof(tmp));
  edges[total].y0 = INT32_MAX; /* sentinel */
  int i = 0;
  edge* e = &edges[0];
  tmp.next = 0;
  if (y_max > h) y_max = h;
  for (int y = e->y0; y < y_max; ++y) {
    edge *last, *prelast, *keep_prelast;
    int sort_flag = 0, draw = 0;
    const int clipline = y < 0;
    prelast = &tmp;
    last = tmp.next;
    while (last || e->y0 == y) {
      if (last && last->y1 == y) { /* the edge ends on this row: drop it */
        prelast->next = last->next;
        last = last->next;
        continue;
      }
      keep_prelast = prelast;
      if (last && (e->y0 > y || last->x < e->x)) { /* next active edge */
        prelast = last;
        last = last->next;
      } else if (i < total) { /* a new edge starts on this row */
        prelast->next = e;
        e->next = last;
        prelast = e;
        e = &edges[++i];
      } else {
        break;
      }
      if (draw) {
        if (!clipline) {
          int x1, x2;
          if (keep_prelast->x > prelast->x) {
            x1 = (int)((prelast->x + XY_ONE - 1) >> XY_SHIFT);
            x2 = (int)(keep_prelast->x >> XY_SHIFT);
          } else {
            x1 = (int)((keep_prelast->x + XY_ONE - 1) >> XY_SHIFT);
            x2 = (int)(prelast->x >> XY_SHIFT);
          }
          if (x1 < w && x2 >= 0) {
            if (x1 < 0) x1 = 0;
            if (x2 >= w) x2 = w - 1;
            for (int x = x1; x <= x2; ++x) img[(size_t)y * w + x] = color;
          }
        }
        keep_prelast->x += keep_prelast->dx;
        prelast->x += prelast->dx;
      }
      draw ^= 1;
    }
    /* keep the active list sorted by x (bubble sort, as upstream) */
    keep_prelast = 0;
    do {
      prelast = &tmp;
      last = tmp.next;
      sort_flag = 0;
      while (last != keep_prelast && last && last->next != 0) {
        edge* te = last->next;
        if (last->x > te->x) {
          prelast->next = te;
          last->next = te->next;
          te->next = last;
          prelast = te;
          sort_flag = 1;
        } else {
          prelast = last;
          last = te;
        }
      }
      keep_prelast = prelast;
    } while (sort_flag && keep_prelast != tmp.next && keep_prelast != &tmp);
  }
}

/* ---- thick lines: cv2.line(..., thickness > 1) = OpenCV ThickLine(flags 3, shift 0): a quad filled by FillConvexPoly in
 * 16.16 fixed point (whose outline is first drawn with Line2), then a filled Circle at both end points.
 * Restated from the published OpenCV 4.1.2 drawing.cpp; PARITY UNPINNED against OpenCV like the rest of this file. */
static void put_point(uint8_t* img, int w, int h, int x, int y, uint8_t color) {
  if (0 <= x && x < w && 0 <= y && y < h) img[(size_t)y * w + x] = color;
}

static void hline(uint8_t* img, int w, int y, int x1, int x2, uint8_t color) {
  for (int x = x1; x <= x2; ++x) img[(size_t)y * w + x] = color;
}

/* OpenCV Line2(): fixed-point DDA between two 16.16 points, one pixel per major-axis step, plus the rounded end point */
static void draw_line2(uint8_t* img, int w, int h, pt64 pt1, pt64 pt2, uint8_t color) {
  if (!clip_line((int64_t)w << XY_SHIFT, (int64_t)h << XY_SHIFT, &pt1, &pt2)) return;
  int64_t dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
  int64_t j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
  int64_t i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
  int64_t x_step, y_step;
  int ecount;
  if (ax > ay) {
    dy = (dy ^ j) - j;
    pt1.x ^= pt2.x & j; pt2.x ^= pt1.x & j; pt1.x ^= pt2.x & j;
    pt1.y ^= pt2.y & j; pt2.y ^= pt1.y & j; pt1.y ^= pt2.y & j;
    x_step = XY_ONE;
    y_step = (dy << XY_SHIFT) / (ax | 1);
    ecount = (int)((pt2.x - pt1.x) >> XY_SHIFT);
  } else {
    dx = (dx ^ i) - i;
    pt1.x ^= pt2.x & i; pt2.x ^= pt1.x & i; pt1.x ^= pt2.x & i;
    pt1.y ^= pt2.y & i; pt2.y ^= pt1.y & i; pt1.y ^= pt2.y & i;
    x_step = (dx << XY_SHIFT) / (ay | 1);
    y_step = XY_ONE;
    ecount = (int)((pt2.y - pt1.y) >> XY_SHIFT);
  }
  pt1.x += (XY_ONE >> 1);
  pt1.y += (XY_ONE >> 1);
  put_point(img, w, h, (int)((pt2.x + (XY_ONE >> 1)) >> XY_SHIFT), (int)((pt2.y + (XY_ONE >> 1)) >> XY_SHIFT), color);
  if (ax > ay) {
    pt1.x >>= XY_SHIFT;
    while (ecount >= 0) {
      put_point(img, w, h, (int)pt1.x, (int)(pt1.y >> XY_SHIFT), color);
      pt1.x++;
      pt1.y += y_step;
      ecount--;
    }
  } else {
    pt1.y >>= XY_SHIFT;
    while (ecount >= 0) {
      put_point(img, w, h, (int)(pt1.x >> XY_SHIFT), (int)pt1.y, color);
      pt1.x += x_step;
      pt1.y++;
      ecount--;
    }
  }
}

/* OpenCV FillConvexPoly(v, npts, color, LINE_8, shift = XY_SHIFT) */
static void fill_convex_poly(uint8_t* img, int w, int h, const pt64* v, int npts, uint8_t color) {
  struct { int idx, di; int64_t x, dx; int ye; } edge[2];
  const int shift = XY_SHIFT;
  const int delta = 1 << shift >> 1;
  int i, y, imin = 0, edges = npts;
  int64_t xmin, xmax, ymin, ymax;
  const int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
  pt64 p0 = v[npts - 1];
  xmin = xmax = v[0].x;
  ymin = ymax = v[0].y;
  for (i = 0; i < npts; i++) {
    pt64 p = v[i];
    if (p.y < ymin) { ymin = p.y; imin = i; }
    if (p.y > ymax) ymax = p.y;
    if (p.x > xmax) xmax = p.x;
    if (p.x < xmin) xmin = p.x;
    draw_line2(img, w, h, p0, p, color);
    p0 = p;
  }
  xmin = (xmin + delta) >> shift;
  xmax = (xmax + delta) >> shift;
  ymin = (ymin + delta) >> shift;
  ymax = (ymax + delta) >> shift;
  if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
  if (ymax > h - 1) ymax = h - 1;
  edge[0].idx = edge[1].idx = imin;
  edge[0].ye = edge[1].ye = y = (int)ymin;
  edge[0].di = 1;
  edge[1].di = npts - 1;
  edge[0].x = edge[1].x = -XY_ONE;
  edge[0].dx = edge[1].dx = 0;
  do {
    for (i = 0; i < 2; i++) {
      if (y >= edge[i].ye) {
        int idx0 = edge[i].idx, di = edge[i].di;
        int idx = idx0 + di;
        if (idx >= npts) idx -= npts;
        int ty = 0;
        for (; edges-- > 0;) {
          ty = (int)((v[idx].y + delta) >> shift);
          if (ty > y) {
            const int64_t xs = v[idx0].x, xe = v[idx].x;
            edge[i].ye = ty;
            edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
            edge[i].x = xs;
            edge[i].idx = idx;
            break;
          }
          idx0 = idx;
          idx += di;
          if (idx >= npts) idx -= npts;
        }
      }
    }
    if (edges < 0) break;
    if (y >= 0) {
      int left = 0, right = 1;
      if (edge[0].x > edge[1].x) { left = 1; right = 0; }
      int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
      int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
      if (xx2 >= 0 && xx1 < w) {
        if (xx1 < 0) xx1 = 0;
        if (xx2 >= w) xx2 = w - 1;
        hline(img, w, y, xx1, xx2, color);
      }
    }
    edge[0].x += edge[0].dx;
    edge[1].x += edge[1].dx;
  } while (++y <= (int)ymax);
}

/* OpenCV Circle(center, radius, color, fill = 1): the midpoint iteration, every step filling four row spans */
static void fill_circle(uint8_t* img, int w, int h, int cx, int cy, int radius, uint8_t color) {
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    int mask;
    const int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
    int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
    if (x11 < w && x12 >= 0 && y21 < h && y22 >= 0) {   /* (the "inside" fast path draws the same pixels) */
      if (x11 < 0) x11 = 0;
      if (x12 > w - 1) x12 = w - 1;
      if ((unsigned)y11 < (unsigned)h) hline(img, w, y11, x11, x12, color);
      if ((unsigned)y12 < (unsigned)h) hline(img, w, y12, x11, x12, color);
      if (x21 < w && x22 >= 0) {
        if (x21 < 0) x21 = 0;
        if (x22 > w - 1) x22 = w - 1;
        if ((unsigned)y21 < (unsigned)h) hline(img, w, y21, x21, x22, color);
        if ((unsigned)y22 < (unsigned)h) hline(img, w, y22, x21, x22, color);
      }
    }
    dy++;
    err += plus;
    plus += 2;
    mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

static int64_t cv_round(double v) { return (int64_t)llrint(v); }   /* cvRound: round half to even (SSE2 cvtsd2si) */

/* OpenCV ThickLine(p0, p1, thickness, LINE_8, flags = 3, shift = 0); thickness <= 1: the thin line above */
static void draw_thick_line(uint8_t* img, int w, int h, int64_t ax, int64_t ay, int64_t bx, int64_t by, uint8_t color,
                            int thickness) {
  if (thickness <= 1) {
    draw_line(img, w, h, ax, ay, bx, by, color);
    return;
  }
  static const double INV_XY_ONE = 1. / XY_ONE;
  pt64 p0 = {ax << XY_SHIFT, ay << XY_SHIFT}, p1 = {bx << XY_SHIFT, by << XY_SHIFT};
  pt64 pt[4], dp = {0, 0};
  const double dx = (p0.x - p1.x) * INV_XY_ONE, dy = (p1.y - p0.y) * INV_XY_ONE;
  double r = dx * dx + dy * dy;
  const int odd = thickness & 1;
  thickness <<= XY_SHIFT - 1;
  if (fabs(r) > 2.220446049250313e-16) {
    r = (thickness + odd * XY_ONE * 0.5) / sqrt(r);
    dp.x = cv_round(dy * r);
    dp.y = cv_round(dx * r);
    pt[0].x = p0.x + dp.x; pt[0].y = p0.y + dp.y;
    pt[1].x = p0.x - dp.x; pt[1].y = p0.y - dp.y;
    pt[2].x = p1.x - dp.x; pt[2].y = p1.y - dp.y;
    pt[3].x = p1.x + dp.x; pt[3].y = p1.y + dp.y;
    fill_convex_poly(img, w, h, pt, 4, color);
  }
  for (int i = 0; i < 2; i++) {
    const int cx = (int)((p0.x + (XY_ONE >> 1)) >> XY_SHIFT), cy = (int)((p0.y + (XY_ONE >> 1)) >> XY_SHIFT);
    fill_circle(img, w, h, cx, cy, (thickness + (XY_ONE >> 1)) >> XY_SHIFT, color);
    p0 = p1;
  }
}

/*
 * Draw list ("commands"), executed in order, per image:
 *   cmds[c] = { kind, a, b, c, plane, color }
 *     kind 0: polygon over body[0..n_body)                      (lib/utils.py:345-355)
 *     kind 1: line joint a -> joint b                           (:357-405)
 *     kind 4: head line, a kind-1 line whose length also enters the throat length (:434-467)
 *     kind 2: neck line (models without head lines): neck = 0.5 * (joint a + joint b) unless a shoulder has a negative
 *             coordinate, drawn to joint c; its length is the throat length (:407-433)
 *     kind 3: face line a -> b, drawn only if its length is < the throat length (:468-505)
 *   Lengths and the neck midpoint are float64 like numpy's (np.linalg.norm = sqrt(dx*dx + dy*dy)).
 * kps: [B][J][2] float (x, y); out: [B][3][H][W] uint8, zero-initialised here.
 */
#include <math.h>

static int joint_ok(const float* k, int j) { return k[2 * j] >= 0.f && k[2 * j + 1] >= 0.f; }

static double seg_len(double ax, double ay, double bx, double by) {
  volatile double dx = ax - bx, dy = ay - by;   /* volatile: no fused multiply-add, every product rounded like numpy's */
  volatile double sx = dx * dx, sy = dy * dy;
  volatile double s = sx + sy;
  return sqrt(s);
}

void stickman_raster_oracle_thick(const float* kps, int B, int J, const int32_t* body, int n_body, const int32_t* cmds,
                                  int n_cmds, uint8_t* out, int H, int W, int thickness);

void stickman_raster_oracle(const float* kps, int B, int J, const int32_t* body, int n_body, const int32_t* cmds,
                            int n_cmds, uint8_t* out, int H, int W) {
  stickman_raster_oracle_thick(kps, B, J, body, n_body, cmds, n_cmds, out, H, W, 1);
}

/* thickness: lib/utils.py:334-339 (img_shape[1] // scale_factor), handed to every cv2.line of the frame */
void stickman_raster_oracle_thick(const float* kps, int B, int J, const int32_t* body, int n_body, const int32_t* cmds,
                                  int n_cmds, uint8_t* out, int H, int W, int thickness) {
  memset(out, 0, (size_t)B * 3 * H * W);
  for (int b = 0; b < B; ++b) {
    const float* k = kps + (size_t)b * J * 2;
    /* throat length first: it only depends on the joints (:415-417, :437-467) */
    double throat = 0.0;
    for (int c = 0; c < n_cmds; ++c) {
      const int32_t* cmd = cmds + 6 * c;
      if (cmd[0] == 4 && joint_ok(k, cmd[1]) && joint_ok(k, cmd[2])) {
        const double l = seg_len(k[2 * cmd[1]], k[2 * cmd[1] + 1], k[2 * cmd[2]], k[2 * cmd[2] + 1]);
        if (l > throat) throat = l;
      } else if (cmd[0] == 2 && joint_ok(k, cmd[1]) && joint_ok(k, cmd[2]) && joint_ok(k, cmd[3])) {
        const double nx = 0.5 * ((double)k[2 * cmd[1]] + (double)k[2 * cmd[2]]);
        const double ny = 0.5 * ((double)k[2 * cmd[1] + 1] + (double)k[2 * cmd[2] + 1]);
        const double l = seg_len(nx, ny, k[2 * cmd[3]], k[2 * cmd[3] + 1]);
        if (l > throat) throat = l;
      }
    }
    for (int c = 0; c < n_cmds; ++c) {
      const int32_t* cmd = cmds + 6 * c;
      uint8_t* plane = out + ((size_t)b * 3 + cmd[4]) * H * W;
      const uint8_t color = (uint8_t)cmd[5];
      if (cmd[0] == 0) {
        int64_t vx[16], vy[16];
        int n = 0;
        if (n_body <= 2) continue;                            /* lib/utils.py:345 */
        for (int i = 0; i < n_body && n < 16; ++i) {
          const float x = k[2 * body[i]], y = k[2 * body[i] + 1];
          if (x >= 0.f && y >= 0.f) { vx[n] = (int64_t)x; vy[n] = (int64_t)y; ++n; }  /* :347-349 */
        }
        if (n > 2) fill_poly(plane, W, H, vx, vy, n, color);                           /* :348 */
      } else if (cmd[0] == 2) {
        if (joint_ok(k, cmd[1]) && joint_ok(k, cmd[2]) && joint_ok(k, cmd[3])) {       /* :408-416 */
          const double nx = 0.5 * ((double)k[2 * cmd[1]] + (double)k[2 * cmd[2]]);
          const double ny = 0.5 * ((double)k[2 * cmd[1] + 1] + (double)k[2 * cmd[2] + 1]);
          draw_thick_line(plane, W, H, (int64_t)nx, (int64_t)ny, (int64_t)k[2 * cmd[3]], (int64_t)k[2 * cmd[3] + 1], color,
                          thickness);
        }
      } else {
        if (!(joint_ok(k, cmd[1]) && joint_ok(k, cmd[2]))) continue;                   /* :358-359 */
        const float ax = k[2 * cmd[1]], ay = k[2 * cmd[1] + 1], bx = k[2 * cmd[2]], by = k[2 * cmd[2] + 1];
        if (cmd[0] == 3 && !(seg_len(ax, ay, bx, by) < throat)) continue;              /* :473-476 */
        draw_thick_line(plane, W, H, (int64_t)ax, (int64_t)ay, (int64_t)bx, (int64_t)by, color, thickness);
      }
    }
  }
}
