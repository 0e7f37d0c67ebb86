// svr_clip.h — what the clipper shares with the setup kernel: the vertex stage of one corner and Sutherland-Hodgman
// against the six planes of the clip volume (contract C2).  The clipper itself runs inside the binning launch
// (k_bin.hip clip_and_bin).
#pragma once
#include "svr_device.h"

namespace svr {

__device__ __forceinline__ void shade_corner(const DrawDesc& d, uint32_t kind, const float* mvp, uint32_t index, VOut& o) {
  VertexRaw v = load_vertex(d.vtx, index);
  if (kind == PIPE_MESH)
    mesh_vert(v, mvp, d.mat, d.color_factors, o);
  else
    colored_triangle_mesh_vert(v, d.mat, o);
}

__device__ __forceinline__ float plane_dist(int plane, const float* c) {
  switch (plane) {
    case 0: return c[3] - c[2];
    case 1: return c[2];
    case 2: return c[3] + c[0];
    case 3: return c[3] - c[0];
    case 4: return c[3] + c[1];
    default: return c[3] - c[1];
  }
}

// Sutherland-Hodgman against the six planes (C2); new vertices always interpolate inside -> outside.
// poly, tmp: room for 12 vertices each (LDS in the clipper: indexed by run-time values they would otherwise live in scratch)
__device__ int clip_polygon(VOut* poly, VOut* tmp, int n) {
  for (int plane = 0; plane < 6 && n >= 3; plane++) {
    int m = 0;
    for (int i = 0; i < n; i++) {
      const VOut& a = poly[i];
      const VOut& b = poly[(i + 1 == n) ? 0 : i + 1];
      float da = plane_dist(plane, a.clip), db = plane_dist(plane, b.clip);
      bool ina = da >= 0.0f, inb = db >= 0.0f;
      if (ina) tmp[m++] = a;
      if (ina != inb) {
        const VOut& pin = ina ? a : b;
        const VOut& pout = ina ? b : a;
        float din = ina ? da : db, dout = ina ? db : da;
        float t = din / (din - dout);
        VOut nv;
        for (int k = 0; k < 4; k++) nv.clip[k] = fmaf(t, pout.clip[k] - pin.clip[k], pin.clip[k]);
        for (int k = 0; k < 8; k++) nv.attr[k] = fmaf(t, pout.attr[k] - pin.attr[k], pin.attr[k]);
        tmp[m++] = nv;
      }
    }
    n = m;
    for (int i = 0; i < n; i++) poly[i] = tmp[i];
  }
  return n >= 3 ? n : 0;
}

}  // namespace svr
