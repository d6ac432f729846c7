// Plan-time coefficient folding + validation: the device-side counterpart of the reference Laplacians'
// __post_init__ methods (gcm_filters/kernels.py:163-170, 259-295, 345-349, 402-406, 454-467, 517-562,
// 630-645) and of the field-independent half of BgridVectorLaplacian.__call__ (kernels.py:746-809, which the
// reference recomputes on every call).  Runs once per plan; the results stay resident in HBM.
//
// Every kernel here works on GLOBAL (ny, nx) planes with periodic wrap in x and y (np.roll semantics) and
// the tripole fold where the grid type has one; row slabs for multi-GPU plans are cut out afterwards.
#include "gcmf_internal.hpp"

#include <cmath>

namespace gcmf {

// validation flag bits written by the kernels
enum : int {
  F_KW_GT1 = 1,       // some kappa_w > 1
  F_KS_GT1 = 2,       // some kappa_s > 1
  F_K_NEAR1 = 4,      // some kappa within 1e-5 of 1
  F_MASK_NONBIN = 8,  // wet mask has a value other than 0 / 1
};

__device__ __forceinline__ int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

#define CELL_LOOP(ny, nx)                                                                             \
  const long long ncell_ = (long long)(ny) * (nx);                                                    \
  for (long long q_ = (long long)blockIdx.x * blockDim.x + threadIdx.x; q_ < ncell_;                  \
       q_ += (long long)gridDim.x * blockDim.x)
#define CELL_JI(nx) const int j = (int)(q_ / (nx)); const int i = (int)(q_ - (long long)j * (nx));
#define AT(p, jj, ii) (p)[(long long)(jj) * nx + (ii)]

// ---- land-mask neighbour bits (REGULAR_WITH_LAND*, TRIPOLAR_REGULAR*) -------------------------------
template <typename T>
__global__ void k_pre_mask(const T *m, uint8_t *bits, int ny, int nx, int tripolar, int *flags) {
  CELL_LOOP(ny, nx) {
    CELL_JI(nx)
    const int ie = wrapi(i + 1, nx), iw = wrapi(i - 1, nx);
    const T mc = AT(m, j, i);
    if (!(mc == T(0) || mc == T(1))) atomicOr(flags, F_MASK_NONBIN);
    unsigned b = (mc != T(0)) ? 1u : 0u;
    if (AT(m, j, ie) != T(0)) b |= 2u;
    if (AT(m, j, iw) != T(0)) b |= 4u;
    T mn, ms;
    if (tripolar) {  // extended mask: row ny is row ny-1 mirrored; row 0's south is that ghost row (kernels.py:461-466)
      mn = (j < ny - 1) ? AT(m, j + 1, i) : AT(m, ny - 1, nx - 1 - i);
      ms = (j > 0) ? AT(m, j - 1, i) : AT(m, ny - 1, nx - 1 - i);
    } else {
      mn = AT(m, wrapi(j + 1, ny), i);
      ms = AT(m, wrapi(j - 1, ny), i);
    }
    if (mn != T(0)) b |= 8u;
    if (ms != T(0)) b |= 16u;
    b |= (unsigned)__popc((b >> 1) & 0xFu) << 5;  // bits 5-7: number of wet neighbours (saves the kernels a popcount)
    bits[q_] = (uint8_t)b;
  }
}

// ---- cells that exchange nothing with their neighbours ---------------------------------------------------
// flux form: all four faces of the cell are closed (and, on the tripole seam, the partner's face onto it)
template <typename T>
__global__ void k_pre_isolated(const T *cE, const T *cN, uint8_t *bits, int ny, int nx, int tripolar, int *count) {
  int mine = 0;
  CELL_LOOP(ny, nx) {
    CELL_JI(nx)
    const int iw = wrapi(i - 1, nx);
    bool open = (AT(cE, j, i) != T(0)) || (AT(cE, j, iw) != T(0)) || (AT(cN, j, i) != T(0));
    if (tripolar) {
      if (j > 0) open = open || (AT(cN, j - 1, i) != T(0));
      if (j == ny - 1) open = open || (AT(cN, ny - 1, nx - 1 - i) != T(0));
    } else {
      open = open || (AT(cN, wrapi(j - 1, ny), i) != T(0));
    }
    bits[q_] = open ? 1u : 0u;
    if (!open) ++mine;
  }
  if (mine) atomicAdd(count, mine);
}
__global__ void k_count_land(const uint8_t *bits, long long n, int *count) {
  int mine = 0;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x)
    mine += (bits[q] & 1u) ? 0 : 1;
  if (mine) atomicAdd(count, mine);
}

// ---- IRREGULAR_WITH_LAND: W/S-face form -> east/north-face coefficient planes ------------------------
// reference: wflux = (g - W g)/dxw*dyw * (m*W m*kappa_w); L = (E wflux - wflux + N sflux - sflux)/area
template <typename T>
__global__ void k_pre_irregular(const T *m, const T *dxw, const T *dyw, const T *dxs, const T *dys, const T *area,
                                const T *kw, const T *ks, T *cE, T *cN, T *ra, int ny, int nx, int *flags) {
  CELL_LOOP(ny, nx) {
    CELL_JI(nx)
    const int ie = wrapi(i + 1, nx), jn = wrapi(j + 1, ny);
    const T kwc = AT(kw, j, i), ksc = AT(ks, j, i);
    int f = 0;
    if (kwc > T(1)) f |= F_KW_GT1;
    if (ksc > T(1)) f |= F_KS_GT1;
    if (fabs((double)kwc - 1.0) <= 1e-5 || fabs((double)ksc - 1.0) <= 1e-5) f |= F_K_NEAR1;
    if (f & ~*(volatile int *)flags) atomicOr(flags, f);
    // east face of (j,i) == west face of (j,i+1)
    cE[q_] = AT(dyw, j, ie) / AT(dxw, j, ie) * (AT(m, j, ie) * AT(m, j, i) * AT(kw, j, ie));
    // north face of (j,i) == south face of (j+1,i)
    cN[q_] = AT(dxs, jn, i) / AT(dys, jn, i) * (AT(m, jn, i) * AT(m, j, i) * AT(ks, jn, i));
    ra[q_] = T(1) / AT(area, j, i);
  }
}

// ---- TRIPOLAR_POP_WITH_LAND: E/N-face form with the north fold ---------------------------------------
template <typename T>
__global__ void k_pre_pop(const T *m, const T *dxe, const T *dye, const T *dxn, const T *dyn, const T *tarea, T *cE,
                          T *cN, T *ra, int ny, int nx) {
  CELL_LOOP(ny, nx) {
    CELL_JI(nx)
    const int ie = wrapi(i + 1, nx);
    const T mc = AT(m, j, i);
    const T mn = (j < ny - 1) ? AT(m, j + 1, i) : AT(m, ny - 1, nx - 1 - i);
    cE[q_] = AT(dye, j, i) / AT(dxe, j, i) * (mc * AT(m, j, ie));
    cN[q_] = AT(dxn, j, i) / AT(dyn, j, i) * (mc * mn);
    ra[q_] = T(1) / AT(tarea, j, i);
  }
}

// ---- MOM5U / MOM5T (kernels.py:345-372, 402-429) ------------------------------------------------------
// Both reduce to north-face / east-face flux form.  NB the reference masks the axis -2 ("fx") difference
// with m*E(m) and the axis -1 ("fy") difference with m*N(m); reproduced as written.
template <typename T>
__global__ void k_pre_mom5(const T *m, const T *dxt, const T *dyt, const T *dxu, const T *dyu, const T *area, T *cE,
                           T *cN, T *ra, int ny, int nx, int is_u) {
  CELL_LOOP(ny, nx) {
    CELL_JI(nx)
    const int ie = wrapi(i + 1, nx), iw = wrapi(i - 1, nx), jn = wrapi(j + 1, ny), js = wrapi(j - 1, ny);
    const T mc = AT(m, j, i);
    const T mask_a = mc * AT(m, j, ie);
    const T mask_b = mc * AT(m, jn, i);
    T cn, ce;
    if (is_u) {
      cn = T(2) / (AT(dxt, jn, i) + AT(dxt, jn, ie)) * mask_a * (T(0.5) * (AT(dyu, j, i) + AT(dyu, jn, i)));
      ce = T(2) / (AT(dyt, j, ie) + AT(dyt, jn, ie)) * mask_b * (T(0.5) * (AT(dxu, j, i) + AT(dxu, j, ie)));
    } else {
      cn = T(2) / (AT(dxu, j, i) + AT(dxu, j, iw)) * mask_a * (T(0.5) * (AT(dyt, j, i) + AT(dyt, jn, i)));
      ce = T(2) / (AT(dyu, j, i) + AT(dyu, js, i)) * mask_b * (T(0.5) * (AT(dxt, j, i) + AT(dxt, j, ie)));
    }
    cN[q_] = cn;
    cE[q_] = ce;
    ra[q_] = T(1) / AT(area, j, i);
  }
}

// ---- VECTOR_C_GRID (kernels.py:630-645 + the field-independent factors of 647-696) -------------------
// planes out: 0 1/dyCu  1 1/dxCu  2 1/dxCv  3 1/dyCv
//             4 a1 = -(k_iso + k_aniso/2) * dyT/dxT*mt * dyT^2     5 a2 = -(..)*dxT/dyT*mt * dyT^2   6 rh = dxT^2/dyT^2
//             7 b1 = -k_iso * dyBu/dxBu*mq * dxBu^2                8 b2 = -k_iso*dxBu/dyBu*mq*dxBu^2  9 rq = dyBu^2/dxBu^2
//             10 rau/dyCu   11 rau/dxCu   12 rav/dyCv   13 rav/dxCv        (rau = area_u>0 ? 1/area_u : 0)
// so that  P = dy2h*str_xx = a1*(u~ - W u~) - a2*(v~ - S v~),  dx2h*str_xx = rh*P,
//          R = dx2q*str_xy = b1*(E v^ - v^) + b2*(N u^ - u^),  dy2q*str_xy = rq*R.
template <typename T> struct CgridIn {
  const T *mt, *mq, *dxT, *dyT, *dxCu, *dyCu, *dxCv, *dyCv, *dxBu, *dyBu, *area_u, *area_v, *kiso, *kaniso;
};
template <typename T> struct Planes14 { T *p[14]; };

template <typename T> __global__ void k_pre_cgrid(const CgridIn<T> in, Planes14<T> o, int ny, int nx) {
  CELL_LOOP(ny, nx) {
    const long long q = q_;
    const T dxT = in.dxT[q], dyT = in.dyT[q], dxBu = in.dxBu[q], dyBu = in.dyBu[q];
    const T dx_dyT = dxT / dyT * in.mt[q], dy_dxT = dyT / dxT * in.mt[q];
    const T dx_dyBu = dxBu / dyBu * in.mq[q], dy_dxBu = dyBu / dxBu * in.mq[q];
    const T dx2h = dxT * dxT, dy2h = dyT * dyT, dx2q = dxBu * dxBu, dy2q = dyBu * dyBu;
    const T au = in.area_u[q], av = in.area_v[q];
    const T rau = au > T(0) ? T(1) / au : T(0), rav = av > T(0) ? T(1) / av : T(0);
    const T kxx = -(in.kiso[q] + T(0.5) * in.kaniso[q]), kxy = -in.kiso[q];
    o.p[0][q] = T(1) / in.dyCu[q];
    o.p[1][q] = T(1) / in.dxCu[q];
    o.p[2][q] = T(1) / in.dxCv[q];
    o.p[3][q] = T(1) / in.dyCv[q];
    o.p[4][q] = kxx * dy_dxT * dy2h;
    o.p[5][q] = kxx * dx_dyT * dy2h;
    o.p[6][q] = dx2h / dy2h;
    o.p[7][q] = kxy * dy_dxBu * dx2q;
    o.p[8][q] = kxy * dx_dyBu * dx2q;
    o.p[9][q] = dy2q / dx2q;
    o.p[10][q] = rau / in.dyCu[q];
    o.p[11][q] = rau / in.dxCu[q];
    o.p[12][q] = rav / in.dyCv[q];
    o.p[13][q] = rav / in.dxCv[q];
  }
}

// ---- VECTOR_B_GRID: the ten POP stencil weights, in the reference's operation order ------------------
// pass 1: kxt = (HTE - N HTE)/TAREA, kyt = (HTN - E HTN)/TAREA       (kernels.py:773, 780)
template <typename T>
__global__ void k_pre_bgrid1(const T *HTE, const T *HTN, const T *TAREA, T *kxt, T *kyt, int ny, int nx) {
  CELL_LOOP(ny, nx) {
    CELL_JI(nx)
    const T rt = T(1) / AT(TAREA, j, i);
    kxt[q_] = (AT(HTE, j, i) - AT(HTE, wrapi(j + 1, ny), i)) * rt;
    kyt[q_] = (AT(HTN, j, i) - AT(HTN, j, wrapi(i + 1, nx))) * rt;
  }
}
// pass 2: planes out: 0 cc  1 DUN  2 DUS  3 DUE  4 DUW  5 DMC  6 DMN  7 DME      (DMS = -DMN, DMW = -DME)
template <typename T> struct BgridIn { const T *DXU, *DYU, *HUS, *HUW, *HTE, *HTN, *UAREA, *kxt, *kyt; };
template <typename T> struct Planes8 { T *p[8]; };

template <typename T> __global__ void k_pre_bgrid2(const BgridIn<T> in, Planes8<T> o, int ny, int nx) {
  CELL_LOOP(ny, nx) {
    CELL_JI(nx)
    const int iw = wrapi(i - 1, nx), js = wrapi(j - 1, ny);
    const T ru = T(1) / AT(in.UAREA, j, i), rdx = T(1) / AT(in.DXU, j, i), rdy = T(1) / AT(in.DYU, j, i);
    auto w1 = [&](int jj, int ii) { return AT(in.HUS, jj, ii) / AT(in.HTE, jj, ii); };
    auto w2 = [&](int jj, int ii) { return AT(in.HUW, jj, ii) / AT(in.HTN, jj, ii); };
    const T dus = w1(j, i) * ru, dun = w1(j, iw) * ru;
    const T duw = w2(j, i) * ru, due = w2(js, i) * ru;
    const T kxu = (AT(in.HUW, js, i) - AT(in.HUW, j, i)) * ru;
    const T kyu = (AT(in.HUS, j, iw) - AT(in.HUS, j, i)) * ru;
    // averages of kxt / kyt: a(jj,ii) = 0.5*(k + W k), b(jj,ii) = 0.5*(k + S k)
    auto ax = [&](const T *k, int jj, int ii) { return T(0.5) * (AT(k, jj, ii) + AT(k, jj, wrapi(ii - 1, nx))); };
    auto ay = [&](const T *k, int jj, int ii) { return T(0.5) * (AT(k, jj, ii) + AT(k, wrapi(jj - 1, ny), ii)); };
    const T dxkx = (ax(in.kxt, js, i) - ax(in.kxt, j, i)) * rdx;
    const T dykx = (ay(in.kxt, j, iw) - ay(in.kxt, j, i)) * rdy;
    const T dyky = (ay(in.kyt, j, iw) - ay(in.kyt, j, i)) * rdy;
    const T dxky = (ax(in.kyt, js, i) - ax(in.kyt, j, i)) * rdx;
    const T dum = -(dxkx + dyky + T(2) * (kxu * kxu + kyu * kyu));
    const T dmc = dxky - dykx;
    const T dme = (T(2) * kyu) / (AT(in.HTN, j, i) + AT(in.HTN, js, i));
    const T dmn = -(T(2) * kxu) / (AT(in.HTE, j, i) + AT(in.HTE, j, iw));
    const T duc = -(dun + dus + due + duw);
    o.p[0][q_] = duc + dum;
    o.p[1][q_] = dun;
    o.p[2][q_] = dus;
    o.p[3][q_] = due;
    o.p[4][q_] = duw;
    o.p[5][q_] = dmc;
    o.p[6][q_] = dmn;
    o.p[7][q_] = dme;
  }
}

// ------------------------------------------------------------------------------------------------------
static int dev_alloc(gcmf_plan *pl, void **p, size_t bytes) {
  GCMF_HIP(hipMalloc(p, bytes));
  pl->owned.push_back(*p);
  return GCMF_OK;
}

// cut rows [first .. first+rows) (mod ny when periodic) of a global plane into a slab allocation
static int cut_rows(gcmf_plan *pl, const void *global_plane, void **slab, size_t elem, hipStream_t s) {
  const int64_t ny = pl->d.ny, nx = pl->d.nx;
  const size_t row_bytes = (size_t)nx * elem;
  int rc = dev_alloc(pl, slab, (size_t)pl->rows_alloc * row_bytes);
  if (rc) return rc;
  const int64_t g0 = pl->d.row_begin - pl->first_owned;
  for (int64_t r = 0; r < pl->rows_alloc;) {  // copy maximal contiguous runs
    int64_t gj = ((g0 + r) % ny + ny) % ny;
    int64_t run = std::min<int64_t>(pl->rows_alloc - r, ny - gj);
    GCMF_HIP(hipMemcpyAsync((char *)*slab + r * row_bytes, (const char *)global_plane + gj * row_bytes, run * row_bytes,
                            hipMemcpyDeviceToDevice, s));
    r += run;
  }
  return GCMF_OK;
}

template <typename T> static int host_rows(const void *dplane, const void *hplane, int64_t row, int64_t nx, std::vector<T> &out) {
  out.resize(nx);
  if (hplane) {
    const T *p = (const T *)hplane + row * nx;
    std::copy(p, p + nx, out.begin());
  } else {
    GCMF_HIP(hipMemcpy(out.data(), (const T *)dplane + row * nx, nx * sizeof(T), hipMemcpyDeviceToHost));
  }
  return GCMF_OK;
}

// tripolar checks of kernels.py:458-459 / 521-522 (southern row all land) and 547-562 (fold of dxn / dyn)
template <typename T>
static int check_tripolar(gcmf_plan *pl, const void *const *dp, const void *const *hp, int mask_idx, int dxn_idx,
                          int dyn_idx) {
  const int64_t ny = pl->d.ny, nx = pl->d.nx;
  std::vector<T> m0, mtop, dxn, dyn;
  int rc = host_rows<T>(dp[mask_idx], hp ? hp[mask_idx] : nullptr, 0, nx, m0);
  if (rc) return rc;
  for (int64_t i = 0; i < nx; ++i)
    if (m0[i] != T(0)) {
      set_error("Wet mask requires zeros in southernmost row");
      return GCMF_ERR_WET_SOUTH_ROW;
    }
  if (dxn_idx < 0) return GCMF_OK;
  if ((rc = host_rows<T>(dp[mask_idx], hp ? hp[mask_idx] : nullptr, ny - 1, nx, mtop))) return rc;
  if ((rc = host_rows<T>(dp[dxn_idx], hp ? hp[dxn_idx] : nullptr, ny - 1, nx, dxn))) return rc;
  if ((rc = host_rows<T>(dp[dyn_idx], hp ? hp[dyn_idx] : nullptr, ny - 1, nx, dyn))) return rc;
  if (nx % 2) {
    set_error("tripolar fold check needs an even nx (the reference raises a broadcast ValueError for odd nx)");
    return GCMF_ERR_INVALID_ARG;
  }
  const int64_t h = nx / 2;
  auto wet_only = [&](const std::vector<T> &a, int64_t i) {  // where(n_wet_mask == 1, a, 0) on row ny-1
    return (mtop[i] * mtop[nx - 1 - i] == T(1)) ? a[i] : T(0);
  };
  for (int64_t i = 0; i < h; ++i) {
    const T left = wet_only(dxn, h - 1 - i), right = wet_only(dxn, h + i);
    if (!(left == right)) {
      set_error("Northernmost row of dxn does not fold onto itself. This is a requirement for using a tripole boundary condition.");
      return GCMF_ERR_DXN_FOLD;
    }
  }
  for (int64_t i = 0; i < h; ++i) {  // np.allclose(a, b): |a-b| <= 1e-8 + 1e-5 |b|, finite only
    const double a = (double)wet_only(dyn, h - 1 - i), b = (double)wet_only(dyn, h + i);
    const bool close = (a == b) || (std::isfinite(a) && std::isfinite(b) && std::fabs(a - b) <= 1e-8 + 1e-5 * std::fabs(b));
    if (!close) {
      set_error("Northernmost row of dyn does not fold onto itself. This is a requirement for using a tripole boundary condition.");
      return GCMF_ERR_DYN_FOLD;
    }
  }
  return GCMF_OK;
}

template <typename T> static int precompute_t(gcmf_plan *pl, const void *const *dp, const void *const *hp) {
  const int ny = (int)pl->d.ny, nx = (int)pl->d.nx;
  const size_t plane = (size_t)ny * nx;
  hipStream_t s = pl->stream;
  const dim3 block(256), grid((unsigned)std::min<size_t>((plane + 255) / 256, 8192));
  Geom &g = pl->g;
  int *dflags = nullptr;
  GCMF_HIP(hipMalloc((void **)&dflags, sizeof(int)));
  struct FlagGuard { int *p; ~FlagGuard() { (void)hipFree(p); } } guard{dflags};
  GCMF_HIP(hipMemsetAsync(dflags, 0, sizeof(int), s));
  auto P = [&](int k) { return (const T *)dp[k]; };

  std::vector<void *> gplanes;  // global coefficient planes produced below (T unless noted)
  std::vector<void *> temps;
  struct TempGuard { std::vector<void *> &v; ~TempGuard() { for (void *p : v) (void)hipFree(p); } } tguard{temps};
  auto galloc = [&](void **p, size_t bytes) -> int {
    GCMF_HIP(hipMalloc(p, bytes));
    temps.push_back(*p);
    return GCMF_OK;
  };
  int rc;
  uint8_t *gbits = nullptr;
  int area_idx = -1;
  const int gt = pl->d.grid_type;

  switch (gt) {
    case GCMF_REGULAR: break;
    case GCMF_REGULAR_AREA_WEIGHTED: area_idx = 0; break;
    case GCMF_REGULAR_WITH_LAND:
    case GCMF_REGULAR_WITH_LAND_AREA_WEIGHTED:
    case GCMF_TRIPOLAR_REGULAR_WITH_LAND_AREA_WEIGHTED: {
      const int mi = (gt == GCMF_REGULAR_WITH_LAND) ? 0 : 1;
      if (gt != GCMF_REGULAR_WITH_LAND) area_idx = 0;
      if (pl->tripolar && (rc = check_tripolar<T>(pl, dp, hp, mi, -1, -1))) return rc;
      if ((rc = galloc((void **)&gbits, plane))) return rc;
      hipLaunchKernelGGL(k_pre_mask<T>, grid, block, 0, s, P(mi), gbits, ny, nx, pl->tripolar ? 1 : 0, dflags);
      break;
    }
    case GCMF_IRREGULAR_WITH_LAND:
    case GCMF_MOM5U:
    case GCMF_MOM5T:
    case GCMF_TRIPOLAR_POP_WITH_LAND: {
      if (gt == GCMF_TRIPOLAR_POP_WITH_LAND && (rc = check_tripolar<T>(pl, dp, hp, 0, 3, 4))) return rc;
      T *c[3];
      for (auto &p : c) {
        if ((rc = galloc((void **)&p, plane * sizeof(T)))) return rc;
        gplanes.push_back(p);
      }
      if (gt == GCMF_IRREGULAR_WITH_LAND)
        hipLaunchKernelGGL(k_pre_irregular<T>, grid, block, 0, s, P(0), P(1), P(2), P(3), P(4), P(5), P(6), P(7), c[0],
                           c[1], c[2], ny, nx, dflags);
      else if (gt == GCMF_TRIPOLAR_POP_WITH_LAND)
        hipLaunchKernelGGL(k_pre_pop<T>, grid, block, 0, s, P(0), P(1), P(2), P(3), P(4), P(5), c[0], c[1], c[2], ny, nx);
      else
        hipLaunchKernelGGL(k_pre_mom5<T>, grid, block, 0, s, P(0), P(1), P(2), P(3), P(4), P(5), c[0], c[1], c[2], ny, nx,
                           gt == GCMF_MOM5U ? 1 : 0);
      break;
    }
    case GCMF_VECTOR_C_GRID: {
      CgridIn<T> in{P(0), P(1), P(2), P(3), P(4), P(5), P(6), P(7), P(8), P(9), P(10), P(11), P(12), P(13)};
      Planes14<T> o;
      for (auto &p : o.p) {
        if ((rc = galloc((void **)&p, plane * sizeof(T)))) return rc;
        gplanes.push_back(p);
      }
      hipLaunchKernelGGL(k_pre_cgrid<T>, grid, block, 0, s, in, o, ny, nx);
      break;
    }
    case GCMF_VECTOR_B_GRID: {
      T *kxt, *kyt;
      if ((rc = galloc((void **)&kxt, plane * sizeof(T)))) return rc;
      if ((rc = galloc((void **)&kyt, plane * sizeof(T)))) return rc;
      hipLaunchKernelGGL(k_pre_bgrid1<T>, grid, block, 0, s, P(4), P(5), P(7), kxt, kyt, ny, nx);
      BgridIn<T> in{P(0), P(1), P(2), P(3), P(4), P(5), P(6), kxt, kyt};
      Planes8<T> o;
      for (auto &p : o.p) {
        if ((rc = galloc((void **)&p, plane * sizeof(T)))) return rc;
        gplanes.push_back(p);
      }
      hipLaunchKernelGGL(k_pre_bgrid2<T>, grid, block, 0, s, in, o, ny, nx);
      break;
    }
    default: set_error("unknown grid type %d", gt); return GCMF_ERR_INVALID_ARG;
  }
  GCMF_HIP(hipGetLastError());

  // scalar plans: which cells never exchange with a neighbour (see gcmf_plan::lbits), in slab-row layout
  int *dcount = nullptr;
  uint8_t *gisol = nullptr;
  const bool flux_kind = (gt == GCMF_IRREGULAR_WITH_LAND || gt == GCMF_MOM5U || gt == GCMF_MOM5T || gt == GCMF_TRIPOLAR_POP_WITH_LAND);
  if (gbits || flux_kind) {
    if ((rc = galloc((void **)&dcount, sizeof(int)))) return rc;
    GCMF_HIP(hipMemsetAsync(dcount, 0, sizeof(int), s));
    if (gbits) {
      hipLaunchKernelGGL(k_count_land, grid, block, 0, s, gbits, (long long)plane, dcount);
    } else {
      if ((rc = galloc((void **)&gisol, plane))) return rc;
      hipLaunchKernelGGL(k_pre_isolated<T>, grid, block, 0, s, (const T *)gplanes[0], (const T *)gplanes[1], gisol, ny, nx,
                         pl->tripolar ? 1 : 0, dcount);
    }
    GCMF_HIP(hipGetLastError());
  }

  // validation results
  int flags = 0;
  GCMF_HIP(hipMemcpyAsync(&flags, dflags, sizeof(int), hipMemcpyDeviceToHost, s));
  GCMF_HIP(hipStreamSynchronize(s));
  if (gt == GCMF_IRREGULAR_WITH_LAND) {
    if (flags & F_KW_GT1) {
      set_error("There are kappa_w values > 1 and this can cause the filter to blow up.Please make sure all kappa_w are <=1.");
      return GCMF_ERR_KAPPA_W_GT1;
    }
    if (flags & F_KS_GT1) {
      set_error("There are kappa_s values > 1 and this can cause the filter to blow up.Please make sure all kappa_s are <=1.");
      return GCMF_ERR_KAPPA_S_GT1;
    }
    if (!(flags & F_K_NEAR1) && !(pl->d.flags & GCMF_PLAN_SKIP_KAPPA_ONE)) {
      set_error("At least one place in the domain must have either kappa_w = 1 or kappa_s = 1. Otherwise the filter's "
                "scale will not be equal to filter_scale anywhere in the domain.");
      return GCMF_ERR_KAPPA_NONE_ONE;
    }
  }
  if (flags & F_MASK_NONBIN) {
    set_error("wet_mask must contain only 0 and 1");
    return GCMF_ERR_UNSUPPORTED;
  }

  // slab cut: coefficient planes live in slab-row layout (== global layout for a single slab)
  for (size_t k = 0; k < gplanes.size(); ++k) {
    void *slab = nullptr;
    if ((rc = cut_rows(pl, gplanes[k], &slab, sizeof(T), s))) return rc;
    g.coef[k] = slab;
  }
  if (gbits) {
    void *slab = nullptr;
    if ((rc = cut_rows(pl, gbits, &slab, 1, s))) return rc;
    g.mbits = (const uint8_t *)slab;
  }
  if (dcount) {
    int n_land = 0;
    GCMF_HIP(hipMemcpyAsync(&n_land, dcount, sizeof(int), hipMemcpyDeviceToHost, s));
    GCMF_HIP(hipStreamSynchronize(s));
    pl->n_land = n_land;  // of the whole grid
    if (gisol) {
      void *slab = nullptr;
      if ((rc = cut_rows(pl, gisol, &slab, 1, s))) return rc;
      pl->lbits = (const uint8_t *)slab;
    } else {
      pl->lbits = g.mbits;
    }
  }
  if (area_idx >= 0) {
    void *slab = nullptr;
    if ((rc = cut_rows(pl, dp[area_idx], &slab, sizeof(T), s))) return rc;
    g.area = slab;
  }
  GCMF_HIP(hipStreamSynchronize(s));
  return GCMF_OK;
}

int precompute(gcmf_plan *pl, const void *const *dplanes, const void *const *hplanes) {
  return pl->d.dtype == GCMF_F64 ? precompute_t<double>(pl, dplanes, hplanes) : precompute_t<float>(pl, dplanes, hplanes);
}

}  // namespace gcmf
