// wstar3d.h — fused T-stage plane-streaming kernel for radius-2 star stencils ("wide
// stars": offsets -2..2 along one axis at a time) on a 3-D (or, with SF_NOJ, 2-D) field
// (CDNA4 / gfx950, wave64).  Compiled at plan creation by hipRTC with the macros and the
// `sf_stage<S>` functors emitted by codegen (codegen.hpp: gen_wide).
//
// Role in the reference: the operators bin/synthesize.py emits for an extent of 2
// (bin/synthesize.py:19-31,41-43,91-104: cross, diffusion), each with the per-point
// semantics of ExpandStencilCPU (stencilflow/stencil/cpu.py:58-115): every out-of-domain
// read of the previous operator's field yields that operator's boundary constant --
// implemented, as in star3d.h, by padding at GLOBAL coordinates at every fused stage.
//
// Decomposition (the star kernel's, widened)
//   block  = tile of TJH x TKH points of the (j,k) plane (whole rows, or -- SF_KTILED --
//            strips with SF_HK halo columns), 2*SF_T halo rows per side, marching along i
//            over one chunk of planes (2*SF_T warm-up planes at each end);
//   thread = SF_RJ consecutive rows x one vector of SF_VK elements (VK = 2 or 4);
//   window = per stage boundary FIVE planes in registers (q-2 .. q+2), rotated by
//            renaming: the step loop is unrolled by five with compile-time slot indices.
//   Step p: stage S produces plane p - 2S from its source window, whose newest plane
//   (input plane p, or what stage S-1 has just produced) sits in slot PH = step mod 5;
//   stage 1 frees input plane p-4 row by row and each freed row at once receives the
//   same row of plane p+1 (a load has a whole step to land).  j-neighbours come from
//   the thread's own rows or -- two rows above / below a thread row -- from the
//   neighbouring thread rows through LDS; k-neighbours from the vector itself or from
//   the two adjacent lanes' vectors (DPP wave_shr / wave_shl), at a wave edge from LDS,
//   where one virtual wave on either side of a row holds the consumer's boundary
//   constant: no thread tests whether a neighbour exists.  Two exchange images
//   alternate, so there is ONE barrier per step.
//
// Macros from codegen: SF_T SF_VK SF_RJ SF_BX SF_BY SF_HK SF_KTILED SF_NOJ SF_N0G SF_N1
//   SF_N2 SF_NJT SF_NKT SF_NT SF_ROW_FENCE SF_OPAQUE SF_KERNEL_NAME; typedef sf_t; struct sf_scalars;
//   struct sf_nb; template<int S> struct sf_stage {bc(), bc_zero, apply()}.

typedef sf_t sf_vec __attribute__((ext_vector_type(SF_VK)));
typedef unsigned sf_u4 __attribute__((ext_vector_type(4)));
typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));

#define SF_R 2
#define SF_W 5  // planes per window
#define SF_REACH (SF_R * SF_T)

#define SF_OOB 0x80000000u
#define SF_PLANE_ELEMS ((long long)SF_N1 * (long long)SF_N2)
#define SF_PLANE_BYTES ((unsigned)(SF_PLANE_ELEMS * (long long)sizeof(sf_t)))
#define SF_RSRC_FLAGS 0x00020000 /* raw buffer, 32-bit data format (gfx9 / CDNA) */

#define SF_TJH (SF_BY * SF_RJ)
#define SF_TKH (SF_BX * SF_VK)
#define SF_WPR (SF_BX / 64)
#if SF_NOJ
#define SF_TJI 1
#else
#define SF_TJI (SF_TJH - 2 * SF_REACH)
#endif
// interior columns of a k-tile: codegen sets SF_TKI when the row is cut into tiles of EQUAL useful width
// (512 columns in three 256-lane tiles: 172 each, instead of 248 + 248 + 16 -- the blocks of a nearly
// empty last tile would take as long as the others while loading nothing)
#ifndef SF_TKI
#define SF_TKI (SF_TKH - 2 * SF_HK)
#endif

// LDS image (one per step parity): per window the first two and the last two rows of
// every thread row, and per row of every thread row the two lowest / two highest
// elements of every wave, with one virtual wave before and one after the row.
#if SF_NOJ
#define SF_ROWS_ELEMS 0
#else
#define SF_ROWS_ELEMS (SF_T * SF_BY * 4 * SF_TKH)
#endif
#define SF_EDGE_WAVES (SF_WPR + 2)
#define SF_EDGE_ELEMS (SF_T * SF_BY * SF_RJ * SF_EDGE_WAVES * 4)
#define SF_IMAGE_ELEMS (SF_ROWS_ELEMS + SF_EDGE_ELEMS)
#define SF_USE_LDS (!(SF_NOJ && SF_WPR == 1))

struct sf_state {
  sf_vec w[SF_T][SF_W][SF_RJ];
};

struct sf_ctx {
  const sf_t* in;
  int tx, ty, lane, wave;
  unsigned jmask, kmask;
  bool tile_inside;  // block-uniform: every point of the tile lies in the (j,k) domain
  int goff, halo, cb, ce;
  int j0, k0;  // global (j, k) of the thread's first point (`copy` boundaries)
  // byte offset of this lane's vector in row r of a plane, or SF_OOB where the lane must
  // not load (outside the (j,k) domain) / store (halo rows and columns)
  unsigned ld_off[SF_RJ], st_off[SF_RJ];
  // LDS element indices that depend on the thread only (everything else of an address is a
  // compile-time offset the instruction carries): the thread's own vector in its thread row's
  // image, in the images of the thread rows above / below, and the edge words of its wave and of
  // the waves before / after it
  int own_rows, above_rows, below_rows, own_edge, lo_edge, hi_edge;
};

// which: 0, 1 = the thread row's first two rows, 2, 3 = its last two
__device__ __forceinline__ constexpr int sf_rows_at(int s, int ty, int which) {
  return ((s * SF_BY + ty) * 4 + which) * SF_TKH;
}
// w = -1 .. SF_WPR (virtual waves at both ends); word 0, 1 = the wave's two lowest elements, 2, 3 = its two highest
__device__ __forceinline__ constexpr int sf_edge_at(int s, int ty, int r, int w, int word) {
  return SF_ROWS_ELEMS + ((((s * SF_BY + ty) * SF_RJ + r) * SF_EDGE_WAVES + (w + 1)) * 4 + word);
}
// the same split into the thread's part (sf_ctx) and a compile-time offset
#define SF_ROWS_CT(s, which) (((s) * SF_BY * 4 + (which)) * SF_TKH)
#define SF_EDGE_CT(s, r, word) ((((s) * SF_BY * SF_RJ + (r)) * SF_EDGE_WAVES) * 4 + (word))

// Value of the adjacent lane through the DPP data path; lanes without a source (lane 0
// when taking from the lower lane, lane 63 from the upper) keep `edge`.
template <bool FROM_LOWER, typename T>
__device__ __forceinline__ T sf_neighbour_lane_or(T x, T edge) {
  constexpr int ctrl = FROM_LOWER ? 0x138 /* wave_shr:1 */ : 0x130 /* wave_shl:1 */;
  if constexpr (sizeof(T) == 4) {
    int moved = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x), ctrl,
                                            0xf, 0xf, false);
    asm volatile("" : "+v"(moved));  // keep LLVM's DPP combiner off it (star3d.h)
    return __builtin_bit_cast(T, moved);
  } else {
    const long long v = __builtin_bit_cast(long long, x), e = __builtin_bit_cast(long long, edge);
    int rlo = __builtin_amdgcn_update_dpp((int)(e & 0xffffffffll), (int)(v & 0xffffffffll), ctrl, 0xf, 0xf, false);
    int rhi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(v >> 32), ctrl, 0xf, 0xf, false);
    asm volatile("" : "+v"(rlo), "+v"(rhi));
    return __builtin_bit_cast(T, ((long long)rhi << 32) | (unsigned int)rlo);
  }
}

// One vector (8, 16 or 32 bytes) through a buffer resource; `off` outside the resource:
// the load returns 0, the store is dropped.
template <typename V, int aux>
__device__ __forceinline__ V sf_buf_load(const __amdgpu_buffer_rsrc_t rs, const unsigned off) {
  if constexpr (sizeof(V) == 8) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, aux));
  } else if constexpr (sizeof(V) == 16) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, aux));
  } else {
    static_assert(sizeof(V) == 32, "vector of 8, 16 or 32 bytes");
    struct { sf_u4 lo, hi; } two;
    two.lo = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, aux);
    two.hi = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, aux);  // (SF_OOB + 16 is outside too)
    return __builtin_bit_cast(V, two);
  }
}
template <typename V, int aux>
__device__ __forceinline__ void sf_buf_store(const V v, const __amdgpu_buffer_rsrc_t rs, const unsigned off) {
  if constexpr (sizeof(V) == 8) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sf_u2, v), rs, off, 0, aux);
  } else if constexpr (sizeof(V) == 16) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sf_u4, v), rs, off, 0, aux);
  } else {
    static_assert(sizeof(V) == 32, "vector of 8, 16 or 32 bytes");
    struct Two { sf_u4 lo, hi; };
    const Two two = __builtin_bit_cast(Two, v);
    __builtin_amdgcn_raw_buffer_store_b128(two.lo, rs, off, 0, aux);
    __builtin_amdgcn_raw_buffer_store_b128(two.hi, rs, off + 16u, 0, aux);
  }
}

// Row r of input plane p (padded with stage 1's boundary constant outside the global
// domain); never under a branch: a plane outside the domain, or a load the caller has
// switched off, gets a resource of zero records.
__device__ __forceinline__ sf_vec sf_load_row(const sf_ctx& cx, const int p, const int r, const bool enabled) {
  const bool plane_ok = enabled && (p + cx.goff >= 0) && (p + cx.goff < SF_N0G);
  const char* base = reinterpret_cast<const char*>(cx.in) + (long long)(p + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0,
                                                                     plane_ok ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
  sf_vec v = sf_buf_load<sf_vec, (SF_NT & 2) ? 2 : 0>(rs, cx.ld_off[r]);
  if constexpr (!sf_stage<1>::bc_zero) {
    const bool ok = plane_ok && cx.ld_off[r] != SF_OOB;
#pragma unroll
    for (int e = 0; e < SF_VK; ++e) v[e] = ok ? v[e] : sf_stage<1>::bc();
  }
  return v;
}

// One stage of the fused group at one step: reads the source window of stage S-1, whose
// newest plane sits in slot PH, and writes plane q = p - 2S of stage S (into its own
// window's slot PH, or to HBM for the last stage).
template <int S, int PH>
__device__ __forceinline__ void sf_stage_step(sf_state& st, const sf_t* lds, const sf_scalars& sc,
                                              sf_t* __restrict__ out, const sf_ctx& cx, const int p,
                                              const bool load_next) {
  constexpr int src = S - 1;
  // source planes q-2, q-1, q, q+1, q+2 (q + 2 is the window's newest plane)
  constexpr int im2 = (PH + 1) % SF_W, im1 = (PH + 2) % SF_W, ic = (PH + 3) % SF_W, ip1 = (PH + 4) % SF_W, ip2 = PH;
  const int q = p - 2 * S;
  const bool plane_in = (q + cx.goff >= 0) && (q + cx.goff < SF_N0G);
  const bool store_plane = (S == SF_T) && q >= cx.cb && q < cx.ce && plane_in;
  sf_t pad = (sf_t)0;
  if constexpr (S < SF_T) pad = sf_stage<(S < SF_T ? S + 1 : S)>::bc();
  // two rows above / below this thread row, from the neighbouring thread rows' images (no test of
  // the thread row: the first / last one reads its own image -- its outer rows are halo rows, what
  // they take as their outer neighbours never reaches a stored value)
  sf_vec above[2], below[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    above[d] = st.w[src][ic][0];
    below[d] = st.w[src][ic][SF_RJ - 1];
  }
  if constexpr (!SF_NOJ) {
#pragma unroll
    for (int d = 0; d < 2; ++d) above[d] = *reinterpret_cast<const sf_vec*>(&lds[cx.above_rows + SF_ROWS_CT(src, 2 + d)]);
  }
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const sf_vec c = st.w[src][ic][r];
    const sf_vec a2 = st.w[src][im2][r], a1 = st.w[src][im1][r], b1 = st.w[src][ip1][r], b2 = st.w[src][ip2][r];
    sf_vec jm2 = c, jm1 = c, jp1 = c, jp2 = c;
    if constexpr (!SF_NOJ) {
      // (the rows below the thread row are fetched when the first row that needs them comes up)
      if (r == (SF_RJ >= 2 ? SF_RJ - 2 : 0)) {
#pragma unroll
        for (int d = 0; d < 2; ++d) below[d] = *reinterpret_cast<const sf_vec*>(&lds[cx.below_rows + SF_ROWS_CT(src, d)]);
      }
      jm2 = (r >= 2) ? st.w[src][ic][r >= 2 ? r - 2 : 0] : above[r];
      jm1 = (r >= 1) ? st.w[src][ic][r >= 1 ? r - 1 : 0] : above[1];
      jp1 = (r + 1 < SF_RJ) ? st.w[src][ic][r + 1 < SF_RJ ? r + 1 : r] : below[0];
      jp2 = (r + 2 < SF_RJ) ? st.w[src][ic][r + 2 < SF_RJ ? r + 2 : r] : below[r + 2 - SF_RJ];
    }
    // the two elements next to this vector on either side: the adjacent lanes' (lane 0 / 63: the
    // neighbouring wave's edge elements -- virtual waves: the boundary constant -- read here, row by
    // row, and handed to the DPP move as its starting destination: no select, no test)
    sf_t e_lo[2], e_hi[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      if constexpr (SF_WPR > 1) {
        e_lo[d] = lds[cx.lo_edge + SF_EDGE_CT(src, r, 2 + d)];
        e_hi[d] = lds[cx.hi_edge + SF_EDGE_CT(src, r, d)];
      } else {
        e_lo[d] = sf_stage<S>::bc();
        e_hi[d] = sf_stage<S>::bc();
      }
    }
    const sf_t lo0 = sf_neighbour_lane_or<true>(c[SF_VK - 2], e_lo[0]);  // k0 - 2
    const sf_t lo1 = sf_neighbour_lane_or<true>(c[SF_VK - 1], e_lo[1]);  // k0 - 1
    const sf_t hi0 = sf_neighbour_lane_or<false>(c[0], e_hi[0]);         // k0 + VK
    const sf_t hi1 = sf_neighbour_lane_or<false>(c[1], e_hi[1]);         // k0 + VK + 1
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) {
      sf_nb nb;
      nb.c = c[v];
      nb.i[0] = a2[v];
      nb.i[1] = a1[v];
      nb.i[2] = b1[v];
      nb.i[3] = b2[v];
      nb.j[0] = jm2[v];
      nb.j[1] = jm1[v];
      nb.j[2] = jp1[v];
      nb.j[3] = jp2[v];
      nb.k[0] = (v >= 2) ? c[v >= 2 ? v - 2 : 0] : (v == 0 ? lo0 : lo1);
      nb.k[1] = (v >= 1) ? c[v >= 1 ? v - 1 : 0] : lo1;
      nb.k[2] = (v + 1 < SF_VK) ? c[v + 1 < SF_VK ? v + 1 : v] : hi0;
      nb.k[3] = (v + 2 < SF_VK) ? c[v + 2 < SF_VK ? v + 2 : v] : (v + 2 == SF_VK ? hi0 : hi1);
      o[v] = sf_stage<S>::apply(nb, sc, v, q + cx.goff, cx.j0 + r, cx.k0 + v);
    }
    if constexpr (S == 1) {
      // row r of the input window's oldest plane (p - 4) is dead now: it receives row r of plane
      // p + 1, which has a whole step to land
      st.w[0][im2][r] = sf_load_row(cx, p + 1, r, load_next);
    }
    if constexpr (S == SF_T) {
      // last stage of the group: interior, in-domain points go to memory (always issued: a plane
      // that is not stored has a resource of zero records, rows and lanes that are not stored an
      // offset outside the plane)
      char* base = reinterpret_cast<char*>(out) + (long long)(q + cx.halo) * (long long)SF_PLANE_BYTES;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(base, 0, store_plane ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
      sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
    } else {
      // pad: outside the global domain the next stage must read ITS constant (skipped by a
      // block-uniform branch for tiles and planes strictly inside)
      if (!(cx.tile_inside && plane_in)) {
        const bool row_in = plane_in && ((cx.jmask >> r) & 1u);
#pragma unroll
        for (int v = 0; v < SF_VK; ++v) o[v] = (row_in && ((cx.kmask >> v) & 1u)) ? o[v] : pad;
      }
      st.w[S < SF_T ? S : 0][PH][r] = o;  // the newest plane of stage S's window
    }
#if SF_ROW_FENCE
    __builtin_amdgcn_sched_barrier(0);  // rows in order: bounds the live temporaries
#endif
  }
}

template <int S, int PH>
__device__ __forceinline__ void sf_later_stages(sf_state& st, const sf_t* lds, const sf_scalars& sc,
                                                sf_t* __restrict__ out, const sf_ctx& cx, const int p) {
  if constexpr (S <= SF_T) {
    sf_stage_step<S, PH>(st, lds, sc, out, cx, p, false);
    sf_later_stages<S + 1, PH>(st, lds, sc, out, cx, p);
  }
}

// the virtual waves' edge words: window s is read by stage s + 1, whose boundary constant
// they hold (never overwritten; the first step's barrier orders them)
template <int S>
__device__ __forceinline__ void sf_edge_prefill(sf_t* lds_all, const sf_ctx& cx) {
  if constexpr (S <= SF_T) {
    if (cx.lane == 0 && (cx.wave == 0 || cx.wave == SF_WPR - 1)) {
#pragma unroll
      for (int image = 0; image < 2; ++image)
#pragma unroll
        for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
          for (int word = 0; word < 4; ++word) {
            sf_t* lds = lds_all + image * SF_IMAGE_ELEMS;
            if (cx.wave == 0) lds[sf_edge_at(S - 1, cx.ty, r, -1, word)] = sf_stage<S>::bc();
            if (cx.wave == SF_WPR - 1) lds[sf_edge_at(S - 1, cx.ty, r, SF_WPR, word)] = sf_stage<S>::bc();
          }
    }
    sf_edge_prefill<S + 1>(lds_all, cx);
  }
}

// One step (newest input plane p) at phase PH.
template <int PH>
__device__ __forceinline__ void sf_step(sf_state& st, sf_t* lds, sf_t* __restrict__ out, const sf_scalars& sc,
                                        const sf_ctx& cx, const int p, const int p_end) {
  constexpr int ic = (PH + 3) % SF_W;
  // Make the windows opaque at the step boundary: where the program's typing turns every neighbour
  // into a double (a float boundary literal) the compiler otherwise keeps the
  // conversions of whole planes alive from one unrolled step to the next -- twice the registers
  // (star3d.h: SF_OPAQUE).  The row in flight (the newest input plane) is left alone.
#if SF_OPAQUE
#pragma unroll
  for (int s = 0; s < SF_T; ++s)
#pragma unroll
    for (int w = 0; w < SF_W; ++w)
#pragma unroll
      for (int r = 0; r < SF_RJ; ++r) {
        if (s == 0 && w == PH) continue;  // loaded during the previous step: no wait here
        asm volatile("" : "+v"(st.w[s][w][r]));
      }
#endif
  // publish what other threads need of every window's centre plane (complete since two steps)
  if constexpr (SF_USE_LDS) {
#pragma unroll
    for (int s = 0; s < SF_T; ++s) {
      if constexpr (!SF_NOJ) {
        *reinterpret_cast<sf_vec*>(&lds[cx.own_rows + SF_ROWS_CT(s, 0)]) = st.w[s][ic][0];
        *reinterpret_cast<sf_vec*>(&lds[cx.own_rows + SF_ROWS_CT(s, 1)]) = st.w[s][ic][SF_RJ > 1 ? 1 : 0];
        *reinterpret_cast<sf_vec*>(&lds[cx.own_rows + SF_ROWS_CT(s, 2)]) = st.w[s][ic][SF_RJ > 1 ? SF_RJ - 2 : 0];
        *reinterpret_cast<sf_vec*>(&lds[cx.own_rows + SF_ROWS_CT(s, 3)]) = st.w[s][ic][SF_RJ - 1];
      }
      if (SF_WPR > 1) {
        if (cx.lane == 0) {
#pragma unroll
          for (int r = 0; r < SF_RJ; ++r) {
            lds[cx.own_edge + SF_EDGE_CT(s, r, 0)] = st.w[s][ic][r][0];
            lds[cx.own_edge + SF_EDGE_CT(s, r, 1)] = st.w[s][ic][r][1];
          }
        }
        if (cx.lane == 63) {
#pragma unroll
          for (int r = 0; r < SF_RJ; ++r) {
            lds[cx.own_edge + SF_EDGE_CT(s, r, 2)] = st.w[s][ic][r][SF_VK - 2];
            lds[cx.own_edge + SF_EDGE_CT(s, r, 3)] = st.w[s][ic][r][SF_VK - 1];
          }
        }
      }
    }
    __syncthreads();
  }
  sf_stage_step<1, PH>(st, lds, sc, out, cx, p, p + 1 < p_end);
  sf_later_stages<2, PH>(st, lds, sc, out, cx, p);
}

// (two waves per SIMD at least: a block of one or two waves -- small grids -- would otherwise be
// compiled for 512 registers per lane and spread into the AGPR file, which the planner reads as a
// slow object)
extern "C" __global__ void __launch_bounds__(SF_BX* SF_BY, 2)
    SF_KERNEL_NAME(const sf_t* __restrict__ in, sf_t* __restrict__ out, sf_scalars sc, sf_auxptrs aux, int halo,
                   int goff, int i_begin, int i_end, int li, int nch1, int i_begin2, int i_end2) {
  (void)aux;
  __shared__ sf_t lds_all[2 * (SF_IMAGE_ELEMS > 0 ? SF_IMAGE_ELEMS : 1)];

  sf_ctx cx;
  cx.in = in;
  cx.tx = threadIdx.x;
  cx.ty = threadIdx.y;
  cx.wave = cx.tx >> 6;
  cx.lane = cx.tx & 63;
  cx.goff = goff;
  cx.halo = halo;
  {
    // (no test of the thread row where a neighbour is read: the first / last one reads its own image)
    const int ta = cx.ty > 0 ? cx.ty - 1 : 0, tb = cx.ty < SF_BY - 1 ? cx.ty + 1 : cx.ty;
    cx.own_rows = sf_rows_at(0, cx.ty, 0) + cx.tx * SF_VK;
    cx.above_rows = sf_rows_at(0, ta, 0) + cx.tx * SF_VK;
    cx.below_rows = sf_rows_at(0, tb, 0) + cx.tx * SF_VK;
    cx.own_edge = sf_edge_at(0, cx.ty, 0, cx.wave, 0);
    cx.lo_edge = sf_edge_at(0, cx.ty, 0, cx.wave - 1, 0);
    cx.hi_edge = sf_edge_at(0, cx.ty, 0, cx.wave + 1, 0);
  }

  // XCD-aware block order: consecutive logical tiles (adjacent in j, sharing halo rows) land on
  // one XCD and therefore one L2 (speed only, never correctness)
  const int nb = gridDim.x, b = blockIdx.x;
  const int xq = nb >> 3, xr = nb & 7, xcd = b & 7;
  const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  const int jt = L % SF_NJT;
  const int kt = (L / SF_NJT) % SF_NKT;
  const int ch = L / (SF_NJT * SF_NKT);

  // chunks [0, nch1) cover planes [i_begin, i_end), later chunks a second range
  if (ch < nch1) {
    cx.cb = i_begin + ch * li;
    cx.ce = (cx.cb + li < i_end) ? cx.cb + li : i_end;
  } else {
    cx.cb = i_begin2 + (ch - nch1) * li;
    cx.ce = (cx.cb + li < i_end2) ? cx.cb + li : i_end2;
  }
  if (cx.cb >= cx.ce) return;

  const int j0 = SF_NOJ ? 0 : (jt * SF_TJI - SF_REACH + cx.ty * SF_RJ);
  const int k0 = SF_KTILED ? (kt * SF_TKI - SF_HK + cx.tx * SF_VK) : cx.tx * SF_VK;
  cx.j0 = j0;
  cx.k0 = k0;
  {
    const int tj0 = SF_NOJ ? 0 : (jt * SF_TJI - SF_REACH);
    const int tk0 = SF_KTILED ? (kt * SF_TKI - SF_HK) : 0;
    cx.tile_inside = tj0 >= 0 && tj0 + (SF_NOJ ? 1 : SF_TJH) <= SF_N1 && tk0 >= 0 && tk0 + SF_TKH <= SF_N2;
  }
  cx.jmask = 0;
  cx.kmask = 0;
  unsigned store_mask = 0;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const int j = j0 + r, tr = cx.ty * SF_RJ + r;
    const bool in_dom = (j >= 0) && (j < SF_N1);
    cx.jmask |= (in_dom ? 1u : 0u) << r;
    store_mask |= ((in_dom && (SF_NOJ || (tr >= SF_REACH && tr < SF_TJH - SF_REACH))) ? 1u : 0u) << r;
  }
#pragma unroll
  for (int v = 0; v < SF_VK; ++v) cx.kmask |= ((k0 + v >= 0 && k0 + v < SF_N2) ? 1u : 0u) << v;
  const bool kvec_in = (cx.kmask & 1u) != 0;  // N2 % VK == 0: whole vector in or out
  bool kload = kvec_in;
  if (SF_KTILED) {
    const int tk = cx.tx * SF_VK;
    if (!(tk >= SF_HK && tk < SF_HK + SF_TKI && kvec_in)) store_mask = 0;
    kload = kload && tk < SF_TKI + 2 * SF_HK;  // lanes beyond the tile's interior and halo have nothing to read
  }
  if (!kvec_in) store_mask = 0;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const unsigned off = (unsigned)(((j0 + r) * SF_N2 + k0) * (int)sizeof(sf_t));
    cx.ld_off[r] = (((cx.jmask >> r) & 1u) && kload) ? off : SF_OOB;
    cx.st_off[r] = ((store_mask >> r) & 1u) ? off : SF_OOB;
  }

  sf_state st;
#pragma unroll
  for (int s = 0; s < SF_T; ++s)
#pragma unroll
    for (int w = 0; w < SF_W; ++w)
#pragma unroll
      for (int r = 0; r < SF_RJ; ++r) st.w[s][w][r] = (sf_vec)(sf_t)0;

  // input planes [p_begin, p_end) are read; step p consumes plane p as its newest one
  const int p_begin = cx.cb - SF_REACH, p_end = cx.ce + SF_REACH;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) st.w[0][0][r] = sf_load_row(cx, p_begin, r, true);  // slot of phase 0
  if constexpr (SF_WPR > 1) sf_edge_prefill<1>(lds_all, cx);
  int image = 0;
  // the trip always runs five steps: up to four surplus steps past p_end compute planes nobody
  // stores (loads and stores are range-guarded), which keeps the loop body free of control flow
  for (int p = p_begin; p < p_end; p += SF_W) {
    sf_step<0>(st, lds_all + image, out, sc, cx, p, p_end);
    image = SF_IMAGE_ELEMS - image;
    sf_step<1>(st, lds_all + image, out, sc, cx, p + 1, p_end);
    image = SF_IMAGE_ELEMS - image;
    sf_step<2>(st, lds_all + image, out, sc, cx, p + 2, p_end);
    image = SF_IMAGE_ELEMS - image;
    sf_step<3>(st, lds_all + image, out, sc, cx, p + 3, p_end);
    image = SF_IMAGE_ELEMS - image;
    sf_step<4>(st, lds_all + image, out, sc, cx, p + 4, p_end);
    image = SF_IMAGE_ELEMS - image;
  }
}
