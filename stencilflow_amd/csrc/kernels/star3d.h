// star3d.h — fused T-stage plane-streaming kernel for radius-1 star stencils on
// a 3-D field (CDNA4 / gfx950, wave64).  Compiled at plan creation by hipRTC
// with the macros and `sf_stage<S>` functors emitted by codegen (codegen.hpp).
//
// Role in the reference: one launch of this kernel evaluates SF_T consecutive
// operators of the chain, each with the per-point semantics of
// ExpandStencilCPU (reference stencilflow/stencil/cpu.py:58-115): every
// out-of-domain read of the previous operator's field yields that operator's
// boundary constant.  That is implemented as padding: at every stage, values at
// coordinates outside the global domain are replaced by the constant the
// *consuming* stage declares, so the stage functor never sees a boundary.
//
// Decomposition
//   block  = tile of TJH x TKH points of the (j,k) plane, halo SF_T rows in j
//            (and SF_HK columns in k when the row is wider than the tile),
//            marching along i over one chunk of planes;
//   thread = SF_RJ consecutive rows x SF_VK consecutive k (one 16-byte vector
//            per row), for every stage: planes q-1 and q in registers.
//   Per step one new input plane is read from HBM (coalesced 16 B/lane) and
//   stage s produces plane p-s; i-neighbours come from the register window,
//   j-neighbours from registers (inner rows) or LDS (first/last row of the
//   adjacent thread row), k-neighbours from the adjacent lane (__shfl) or,
//   at a wave edge, from LDS.
//
// Macros from codegen: SF_T SF_VK SF_RJ SF_BX SF_BY SF_HK SF_KTILED
//   SF_N0G SF_N1 SF_N2 SF_NJT SF_NKT SF_KERNEL_NAME, typedef sf_t,
//   struct sf_scalars, template<int S> struct sf_stage {bc(), apply()}.

typedef sf_t sf_vec __attribute__((ext_vector_type(SF_VK)));

#define SF_TJH (SF_BY * SF_RJ)
#define SF_TKH (SF_BX * SF_VK)
#define SF_WPR (SF_BX / 64)
#if SF_NOJ
#define SF_TJI 1  // 2-D programs: no tiled row axis, the stream axis is j
#else
#define SF_TJI (SF_TJH - 2 * SF_T)
#endif
#define SF_TKI (SF_TKH - 2 * SF_HK)

// LDS image: per stage the first and last row of every thread row, plus the
// first / last column element of every wave for each of its rows.
#if SF_NOJ
#define SF_ROWS_ELEMS 0
#else
#define SF_ROWS_ELEMS (SF_T * SF_BY * 2 * SF_TKH)
#endif
#define SF_USE_LDS (!(SF_NOJ && SF_WPR == 1))
#define SF_EDGE_ELEMS (SF_T * SF_BY * SF_RJ * SF_WPR * 2)

struct sf_state {
  sf_vec prev[SF_T][SF_RJ];
  sf_vec cur[SF_T][SF_RJ];
};

__device__ __forceinline__ int sf_rows_at(int s, int ty, int which) {
  return ((s * SF_BY + ty) * 2 + which) * SF_TKH;
}
__device__ __forceinline__ int sf_edge_at(int s, int ty, int r, int w, int side) {
  return SF_ROWS_ELEMS + ((((s * SF_BY + ty) * SF_RJ + r) * SF_WPR + w) * 2 + side);
}

// One stage of the fused group at one step.  `fresh` holds plane q+1 of the
// source field (stage S-1) on entry and plane q of stage S on return; the
// source window is rotated row by row so that at most three planes per stage
// boundary are live.
template <int S>
__device__ __forceinline__ void sf_stage_step(
    sf_state& st, sf_vec (&fresh)[SF_RJ], const sf_t* lds, const sf_scalars& sc,
    sf_t* __restrict__ out, const int tx, const int ty, const int lane,
    const int wave, const unsigned jmask, const unsigned kmask,
    const unsigned store_mask, const int p, const int goff, const int halo,
    const int cb, const int ce, const int j0, const int k0) {
  constexpr int src = S - 1;
  // first / last row of the neighbouring thread rows (LDS)
  sf_vec jm0 = st.cur[src][0], jpl = st.cur[src][SF_RJ - 1];
  if constexpr (!SF_NOJ) {
    if (ty > 0)
      jm0 = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(src, ty - 1, 1) + tx * SF_VK]);
    if (ty < SF_BY - 1)
      jpl = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(src, ty + 1, 0) + tx * SF_VK]);
  }
  const int q = p - S;  // plane this stage produces (local owned coords)
  const bool plane_in = (q + goff >= 0) && (q + goff < SF_N0G);
  const bool store_plane = (S == SF_T) && q >= cb && q < ce && plane_in;
  sf_t pad = (sf_t)0;
  if constexpr (S < SF_T) pad = sf_stage<(S < SF_T ? S + 1 : S)>::bc();
  sf_vec jm = jm0;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const sf_vec c = st.cur[src][r];
    const sf_vec im = st.prev[src][r];
    const sf_vec ip = fresh[r];
    const sf_vec jp = (r < SF_RJ - 1) ? st.cur[src][r < SF_RJ - 1 ? r + 1 : r] : jpl;
    // innermost-dimension halo: adjacent lanes hold the adjacent vectors
    sf_t km_e = __shfl_up(c[SF_VK - 1], 1);
    sf_t kp_e = __shfl_down(c[0], 1);
    if (lane == 0)
      km_e = (SF_WPR > 1 && wave > 0) ? lds[sf_edge_at(src, ty, r, wave > 0 ? wave - 1 : 0, 1)]
                                      : sf_stage<S>::bc();
    if (lane == 63)
      kp_e = (SF_WPR > 1 && wave < SF_WPR - 1)
                 ? lds[sf_edge_at(src, ty, r, wave < SF_WPR - 1 ? wave + 1 : wave, 0)]
                 : sf_stage<S>::bc();
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) {
      const sf_t km = (v > 0) ? c[v > 0 ? v - 1 : 0] : km_e;
      const sf_t kp = (v < SF_VK - 1) ? c[v < SF_VK - 1 ? v + 1 : v] : kp_e;
      o[v] = sf_stage<S>::apply(c[v], im[v], ip[v], jm[v], jp[v], km, kp, sc);
#if SF_ROW_FENCE > 1
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // rotate this row of the source window: plane q+1 becomes "current"
    jm = c;
    st.prev[src][r] = c;
    st.cur[src][r] = ip;
    if constexpr (S == SF_T) {
      // last stage of the group: write interior, in-domain points
      if (store_plane && ((store_mask >> r) & 1u)) {
        // wave-uniform plane base (SGPR pair) + 32-bit in-plane offset
        sf_t* plane = out + (size_t)(q + halo) * ((size_t)SF_N1 * SF_N2);
        *reinterpret_cast<sf_vec*>(plane + (unsigned)((j0 + r) * SF_N2 + k0)) = o;
      }
    } else {
      // pad: outside the global domain the next stage must read ITS constant
      const bool row_in = plane_in && ((jmask >> r) & 1u);
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) o[v] = (row_in && ((kmask >> v) & 1u)) ? o[v] : pad;
      fresh[r] = o;
    }
#if SF_ROW_FENCE
    __builtin_amdgcn_sched_barrier(0);  // rows in order: bounds the live f64 temporaries
#endif
  }
}

template <int S>
__device__ __forceinline__ void sf_later_stages(
    sf_state& st, sf_vec (&fresh)[SF_RJ], const sf_t* lds, const sf_scalars& sc,
    sf_t* __restrict__ out, const int tx, const int ty, const int lane,
    const int wave, const unsigned jmask, const unsigned kmask,
    const unsigned store_mask, const int p, const int goff, const int halo,
    const int cb, const int ce, const int j0, const int k0) {
  if constexpr (S <= SF_T) {
    sf_stage_step<S>(st, fresh, lds, sc, out, tx, ty, lane, wave, jmask, kmask, store_mask, p,
                     goff, halo, cb, ce, j0, k0);
    sf_later_stages<S + 1>(st, fresh, lds, sc, out, tx, ty, lane, wave, jmask, kmask,
                           store_mask, p, goff, halo, cb, ce, j0, k0);
  }
}

extern "C" __global__ void __launch_bounds__(SF_BX* SF_BY)
    SF_KERNEL_NAME(const sf_t* __restrict__ in, sf_t* __restrict__ out, sf_scalars sc,
                   int halo, int goff, int i_begin, int i_end, int li) {
  // SF_LDS_DB: two exchange images used alternately -> one barrier per step
  __shared__ sf_t lds_all[(SF_LDS_DB ? 2 : 1) * (SF_ROWS_ELEMS + SF_EDGE_ELEMS)];

  const int tx = threadIdx.x, ty = threadIdx.y;
  const int lane = tx & 63, wave = tx >> 6;

  // XCD-aware block order: consecutive logical tiles (adjacent in j, sharing
  // halo rows) land on one XCD and therefore one L2 (blocks are dealt
  // round-robin over the 8 XCDs; speed only, never correctness).
  const int nb = gridDim.x, b = blockIdx.x;
  const int xq = nb >> 3, xr = nb & 7, xcd = b & 7;
  const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  const int jt = L % SF_NJT;
  const int kt = (L / SF_NJT) % SF_NKT;
  const int ch = L / (SF_NJT * SF_NKT);

  const int cb = i_begin + ch * li;
  const int ce = (cb + li < i_end) ? cb + li : i_end;
  if (cb >= ce) return;

  const int j0 = SF_NOJ ? 0 : (jt * SF_TJI - SF_T + ty * SF_RJ);
  const int k0 = SF_KTILED ? (kt * SF_TKI - SF_HK + tx * SF_VK) : tx * SF_VK;

  unsigned jmask = 0, kmask = 0, store_mask = 0;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const int j = j0 + r, tr = ty * SF_RJ + r;
    const bool in = (j >= 0) && (j < SF_N1);
    jmask |= (in ? 1u : 0u) << r;
    store_mask |= ((in && (SF_NOJ || (tr >= SF_T && tr < SF_TJH - SF_T))) ? 1u : 0u) << r;
  }
#pragma unroll
  for (int v = 0; v < SF_VK; ++v)
    kmask |= ((k0 + v >= 0 && k0 + v < SF_N2) ? 1u : 0u) << v;
  const bool kvec_in = (kmask & 1u) != 0;  // N2 % VK == 0: whole vector in or out
  if (SF_KTILED) {
    const int tk = tx * SF_VK;
    if (!(tk >= SF_HK && tk < SF_TKH - SF_HK && kvec_in)) store_mask = 0;
  }

  sf_state st;
#pragma unroll
  for (int s = 0; s < SF_T; ++s)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r) {
      st.prev[s][r] = (sf_vec)(sf_t)0;
      st.cur[s][r] = (sf_vec)(sf_t)0;
    }

  const sf_t pad0 = sf_stage<1>::bc();
  auto load_plane = [&](int p, sf_vec(&dst)[SF_RJ]) {
    const bool plane_in = (p + goff >= 0) && (p + goff < SF_N0G);
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r) {
      sf_vec v = (sf_vec)pad0;
      if (plane_in && ((jmask >> r) & 1u) && kvec_in) {
        const sf_t* plane = in + (size_t)(p + halo) * ((size_t)SF_N1 * SF_N2);
        v = *reinterpret_cast<const sf_vec*>(plane + (unsigned)((j0 + r) * SF_N2 + k0));
      }
      dst[r] = v;
    }
  };

  sf_vec pre[SF_RJ];
  load_plane(cb - SF_T, pre);

  int parity = 0;
  for (int p = cb - SF_T; p < ce + SF_T; ++p) {
    sf_t* lds = lds_all + (SF_LDS_DB ? parity * (SF_ROWS_ELEMS + SF_EDGE_ELEMS) : 0);
    parity ^= 1;
    // publish the rows / columns other threads need of every stage's current plane
#pragma unroll
    for (int s = 0; s < SF_T; ++s) {
      if constexpr (!SF_NOJ) {
        *reinterpret_cast<sf_vec*>(&lds[sf_rows_at(s, ty, 0) + tx * SF_VK]) = st.cur[s][0];
        *reinterpret_cast<sf_vec*>(&lds[sf_rows_at(s, ty, 1) + tx * SF_VK]) = st.cur[s][SF_RJ - 1];
      }
      if (SF_WPR > 1) {
        if (lane == 0) {
#pragma unroll
          for (int r = 0; r < SF_RJ; ++r) lds[sf_edge_at(s, ty, r, wave, 0)] = st.cur[s][r][0];
        }
        if (lane == 63) {
#pragma unroll
          for (int r = 0; r < SF_RJ; ++r)
            lds[sf_edge_at(s, ty, r, wave, 1)] = st.cur[s][r][SF_VK - 1];
        }
      }
    }
    if (SF_USE_LDS) __syncthreads();
    // stage 1 consumes the prefetched input plane p ...
    sf_stage_step<1>(st, pre, lds, sc, out, tx, ty, lane, wave, jmask, kmask, store_mask, p, goff,
                     halo, cb, ce, j0, k0);
    sf_vec fresh[SF_RJ];
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r) fresh[r] = pre[r];
    // ... whose registers then receive input plane p+1 while later stages run
    if (p + 1 < ce + SF_T) load_plane(p + 1, pre);
    sf_later_stages<2>(st, fresh, lds, sc, out, tx, ty, lane, wave, jmask, kmask, store_mask, p,
                       goff, halo, cb, ce, j0, k0);
    if (SF_USE_LDS && !SF_LDS_DB) __syncthreads();
  }
}
