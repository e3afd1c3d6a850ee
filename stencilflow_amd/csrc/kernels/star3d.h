// star3d.h — fused T-stage plane-streaming kernel for radius-1 star stencils on
// a 3-D (or, with SF_NOJ, 2-D) field (CDNA4 / gfx950, wave64).  Compiled at plan
// creation by hipRTC with the macros and `sf_stage<S>` functors emitted by
// codegen (codegen.hpp).
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
//            per row); for every stage boundary a window of three planes
//            (q-1, q, q+1) lives in registers.
//   Input planes: the plane for the next step is loaded into the window slot stage 1 has just freed, or (SF_RING4:
//   3-D float32 chains and every 2-D program) the input window has FOUR slots rotating with period 4 -- the step loop is
//   unrolled by 4 -- so the plane after next is loaded straight into the slot freed now: two steps to land, no copy.
//   Per step one new input plane is read from HBM (coalesced 16 B/lane);
//   i-neighbours come from the register window, j-neighbours from registers
//   (inner rows) or LDS (first/last row of the adjacent thread row),
//   k-neighbours from the adjacent lane (DPP wave_shr/wave_shl) or, at a wave
//   edge, from LDS.
//   The three window slots rotate by *phase*: the step loop is unrolled by 3
//   with compile-time slot indices, so no register is ever copied -- the slot
//   that held plane q-1 receives plane q+2.
//
// Step order
//   3-D: stage 1 first, then stages 2..T, each consuming what the previous one produced in the same step; stage S
//   produces plane p-S.
//   2-D (a block is a lone wave; chains only): the storing stage T first, stage 1 last -- every stage reads only planes
//   finished in earlier steps (stage S produces plane p-(2S-1)), so the stages of one step are independent of each other.
//
// Macros from codegen: SF_T SF_VK SF_RJ SF_BX SF_BY SF_HK SF_KTILED SF_NOJ
//   SF_N0G SF_N1 SF_N2 SF_NJT SF_NKT SF_ROW_FENCE SF_OPAQUE SF_RING4 SF_AUX_AHEAD SF_NT SF_KERNEL_NAME
//   [SF_NS SF_NW SF_NOUT: DAG groups; SF_AUX_PASS];
//   typedef sf_t, struct sf_scalars, struct sf_auxptrs,
//   template<int S> struct sf_stage {bc(), bc_zero, bc_copy, load_aux(), apply()}.

// A fused group is a DAG of SF_NS stages over SF_NW register windows with SF_NOUT materialised fields
// (codegen: sf_stage<S>::src / src2 / dst / out / depth / refill, sf_win<W>::bc); SF_T is its depth, i.e. the halo
// of a tile and the warm-up planes of a chunk.  A chain: SF_NS = SF_NW = SF_T, SF_NOUT = 1.
#ifndef SF_NS
#define SF_NS SF_T
#endif
#ifndef SF_NW
#define SF_NW SF_T
#endif
#ifndef SF_NOUT
#define SF_NOUT 1
#endif
#ifndef SF_DAG
#define SF_DAG 0
#endif
typedef sf_t sf_vec __attribute__((ext_vector_type(SF_VK)));
// the fields a launch materialises (argument `out` is p[0]; further outputs of a DAG group follow in `more`)
struct sf_outptrs {
  sf_t* p[SF_NOUT > 0 ? SF_NOUT : 1];
};
struct sf_more_outs {
  void* p[SF_NOUT > 1 ? SF_NOUT - 1 : 1];
};

#ifndef SF_AUX_AHEAD
#define SF_AUX_AHEAD 0
#endif
// SF_SKIP_ROWS: a stage does not evaluate the rows of the tile's halo that no later stage reads.  Stage S of T
// is read back through S' > S stages of reach one row each, so of the SF_TJH rows of a tile only rows
// [S, SF_TJH - S) of its output are ever used (the storing stage: exactly the rows it stores).  The test is on
// the thread row, which is the same for all lanes of a wave: a scalar branch around the row's arithmetic.  Only the
// first and last thread rows of a block skip anything -- with 2 waves per thread row and 4 thread rows each SIMD holds
// one such wave, so every SIMD issues (RJ - S) instead of RJ rows per stage for one of its two waves.
// (diagnostic values: 2 = buffer loads only, 3 = buffer stores only)

// SF_BUFFER_IO: planes are read and written with buffer instructions whose
// resource describes exactly one plane.  A lane (or a whole row, or -- with zero
// records -- a whole plane) that must not touch memory is given an offset
// outside the resource: the hardware drops such stores and returns 0 for such
// loads.  No vector-memory instruction of the step loop sits under a branch any
// more, so the compiler can count them and waits for the loads it needs
// (`s_waitcnt vmcnt(N)`) instead of for everything in flight, stores included
// (`vmcnt(0)`, once per step, as soon as any of them is conditional).
typedef unsigned sf_u4 __attribute__((ext_vector_type(4)));
#define SF_OOB 0x80000000u
#define SF_PLANE_ELEMS ((long long)SF_N1 * (long long)SF_N2)
#define SF_PLANE_BYTES ((unsigned)(SF_PLANE_ELEMS * (long long)sizeof(sf_t)))
#define SF_RSRC_FLAGS 0x00020000 /* raw buffer, 32-bit data format (gfx9 / CDNA) */
typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));


// window slots per stage and period of the phase rotation
// (SF_RING4: the four-slot input ring; SF_INFLIGHT the planes in flight beside the window)
#define SF_INFLIGHT (SF_RING4 ? 1 : 0)
#define SF_SLOTS (3 + SF_INFLIGHT)

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
#define SF_ROWS_ELEMS (SF_NW * SF_BY * 2 * SF_TKH)
#endif
#define SF_USE_LDS (!(SF_NOJ && SF_WPR == 1))
// (one virtual wave below and one above every row hold the boundary
// constant, so a wave reads its neighbours' edge columns without testing whether
// they exist)
#define SF_VWAVES (SF_WPR > 1)
#define SF_EDGE_WAVES (SF_VWAVES ? SF_WPR + 2 : SF_WPR)
#define SF_EDGE_ELEMS (SF_NW * SF_BY * SF_RJ * SF_EDGE_WAVES * 2)
#define SF_IMAGE_ELEMS (SF_ROWS_ELEMS + SF_EDGE_ELEMS)

// w[s][slot][row]: planes of stage-s data (s = 0 is the input field).  At
// phase PH the slots hold  prev = PH % 3,  cur = (PH + 1) % 3,  next = (PH + 2) % 3
// (SF_RING4: modulo 4; slot (PH + 3) % 4 of the input window holds the plane in
// flight, the same slot of the later stages' windows is never live).
// (SF_AUX_AHEAD 2) one row set of auxiliary values per stage, requested a step ahead
#ifndef SF_AUX_CARRY
#define SF_AUX_CARRY 0
#endif
template <int N>
struct sf_auxslots : sf_auxslots<N - 1> {
  typename sf_stage<N>::aux_row a[SF_RJ];
#if SF_AUX_CARRY
  typename sf_stage<N>::aux_row used[SF_RJ];  // the rows this stage used in the current step (for stage N + 1)
#endif
};
template <>
struct sf_auxslots<0> {};

struct sf_state
#if SF_AUX_AHEAD == 2
    : sf_auxslots<SF_NS>
#endif
{
  sf_vec w[SF_NW][SF_SLOTS][SF_RJ];
};

#ifndef SF_AUX_PASS
#define SF_AUX_PASS 0
#endif
#if SF_AUX_PASS
typedef sf_stage<2>::aux_row sf_aux_passed;
static_assert(sizeof(sf_stage<1>::aux_row) == sizeof(sf_aux_passed), "stages must share the auxiliary row layout");
#endif

struct sf_ctx {
  const sf_t* in;
  sf_auxptrs aux;  // centre-only auxiliary fields (same layout as `in`)
#if SF_AUX_PASS
  sf_aux_passed* aux_lds;  // [step parity][row][thread]: stage 1's auxiliary rows on their way to stage 2
#endif
  int tx, ty, lane, wave;
  unsigned jmask, kmask, store_mask;
  bool kvec_in;
  bool tile_inside;  // block-uniform: every point of the tile lies in the (j,k) domain
  int goff, halo, cb, ce, j0, k0;
  // byte offset of this lane's vector in row r of a plane, or SF_OOB where the
  // lane must not load (outside the (j,k) domain) / store (halo rows and columns)
  unsigned ld_off[SF_RJ], st_off[SF_RJ];
#if SF_NT & 4
  unsigned nt_rows;  // wave-uniform: bit r set = row r of this thread row is read by this block only
#endif
};

__device__ __forceinline__ int sf_rows_at(int s, int ty, int which) {
  return ((s * SF_BY + ty) * 2 + which) * SF_TKH;
}
__device__ __forceinline__ int sf_edge_at(int s, int ty, int r, int w, int side) {
  return SF_ROWS_ELEMS +
         ((((s * SF_BY + ty) * SF_RJ + r) * SF_EDGE_WAVES + (SF_VWAVES ? w + 1 : w)) * 2 + side);
}

// Value of the adjacent lane (lane-1 for DOWN = false ... see callers) through the
// DPP data path (v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx9): one VALU move,
// no trip through the LDS crossbar that __shfl_up/__shfl_down (ds_bpermute) take.
template <bool FROM_LOWER, typename T>
__device__ __forceinline__ T sf_neighbour_lane(T x) {
  constexpr int ctrl = FROM_LOWER ? 0x138 /* wave_shr:1 */ : 0x130 /* wave_shl:1 */;
  if constexpr (sizeof(T) == 4) {
    const int v = __builtin_bit_cast(int, x);
    // bound_ctrl: lanes without a source read 0 and nothing of the old value is
    // kept, so no copy precedes the DPP move (the caller replaces lane 0 / 63)
    // (the result is made opaque: LLVM's DPP combiner otherwise folds the move into
    // a consuming f64 conversion, which gfx950 cannot encode with wave_shr/wave_shl)
    int moved = __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
    asm volatile("" : "+v"(moved));
    return __builtin_bit_cast(T, moved);
  } else {
    const long long v = __builtin_bit_cast(long long, x);
    const int lo = (int)(v & 0xffffffffll), hi = (int)(v >> 32);
    int rlo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, true);
    int rhi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, true);
    asm volatile("" : "+v"(rlo), "+v"(rhi));
    return __builtin_bit_cast(T, ((long long)rhi << 32) | (unsigned int)rlo);
  }
}

// As above, but lanes without a source (lane 0 when taking from the lower lane,
// lane 63 from the upper) keep `edge`: one DPP move whose destination register
// starts out as the value for the wave's edge lane -- no select afterwards.
template <bool FROM_LOWER, typename T>
__device__ __forceinline__ T sf_neighbour_lane_or(T x, T edge) {
  constexpr int ctrl = FROM_LOWER ? 0x138 /* wave_shr:1 */ : 0x130 /* wave_shl:1 */;
  if constexpr (sizeof(T) == 4) {
    int moved = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x), ctrl,
                                            0xf, 0xf, false);
    asm volatile("" : "+v"(moved));  // keep LLVM's DPP combiner off it (see above)
    return __builtin_bit_cast(T, moved);
  } else {
    const long long v = __builtin_bit_cast(long long, x), e = __builtin_bit_cast(long long, edge);
    int rlo = __builtin_amdgcn_update_dpp((int)(e & 0xffffffffll), (int)(v & 0xffffffffll), ctrl, 0xf, 0xf, false);
    int rhi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(v >> 32), ctrl, 0xf, 0xf, false);
    asm volatile("" : "+v"(rlo), "+v"(rhi));
    return __builtin_bit_cast(T, ((long long)rhi << 32) | (unsigned int)rlo);
  }
}

// One vector (4, 8, 16 or 32 bytes: 1 float / 2 floats or 1 double / 4 floats or 2 doubles / 4 doubles) through a
// buffer resource; `off` outside the resource: the load returns 0, the store is dropped.
template <typename V, int aux>
__device__ __forceinline__ V sf_buf_load(const __amdgpu_buffer_rsrc_t rs, const unsigned off) {
  if constexpr (sizeof(V) == 4) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, aux));
  } else if constexpr (sizeof(V) == 8) {
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
  if constexpr (sizeof(V) == 4) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, aux);
  } else if constexpr (sizeof(V) == 8) {
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

// Is row r of input plane p inside the global domain (and inside what this chunk reads)?
__device__ __forceinline__ bool sf_row_ok(const sf_ctx& cx, const int p, const int r) {
  const bool plane_in = (p + cx.goff >= 0) && (p + cx.goff < SF_N0G);
  return plane_in && ((cx.jmask >> r) & 1u) && cx.kvec_in;
}

// Row r of input plane p (padded with stage 1's boundary constant outside the
// global domain).
__device__ __forceinline__ sf_vec sf_load_row(const sf_ctx& cx, const int p, const int r,
                                              const bool enabled = true) {
  // wave-uniform: a plane outside the global domain (or a load the caller has
  // switched off) gets a resource of zero records
  const bool plane_ok =
      enabled && (p + cx.goff >= 0) && (p + cx.goff < SF_N0G);
  const char* base = reinterpret_cast<const char*>(cx.in) + (long long)(p + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(base), 0, plane_ok ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
#if SF_NT & 4
  // rows of the tile no other block reads (neither a neighbouring tile's halo nor its source of halo rows) are
  // streamed -- non-temporal, as a flat copy's loads would be -- while the shared rows keep the default policy
  // and meet their second reader in the XCD's L2 (wave-uniform choice: a scalar branch, one load either way)
  sf_vec v;
  if ((cx.nt_rows >> r) & 1u) v = sf_buf_load<sf_vec, 2>(rs, cx.ld_off[r]);
  else v = sf_buf_load<sf_vec, 0>(rs, cx.ld_off[r]);
#else
  sf_vec v = sf_buf_load<sf_vec, (SF_NT & 2) ? 2 : 0>(rs, cx.ld_off[r]);
#endif
  if constexpr (!sf_win<0>::bc_zero) {
    const bool ok = plane_ok && cx.ld_off[r] != SF_OOB;
#pragma unroll
    for (int e = 0; e < SF_VK; ++e) v[e] = ok ? v[e] : sf_win<0>::bc();
  }
  return v;
}

// `dst = row r of plane p` if `cond` (wave-uniform).  SF_BUFFER_IO: always issued,
// with a resource of zero records when `cond` is false (dst then holds the padding).
#define SF_LOAD_ROW_IF(cond, dst, p, r) dst = sf_load_row(cx, p, r, cond)

// Auxiliary values of stage S for row r of plane q: only planes this chunk's
// stage S really evaluates are touched (the surplus steps of the last trip and
// the warm-up steps must not reach outside the buffer).
template <int S>
__device__ __forceinline__ typename sf_stage<S>::aux_row sf_aux_row(const sf_ctx& cx, const int q, const int r) {
  const bool plane_ok = (q + cx.goff >= 0) && (q + cx.goff < SF_N0G) && (q + cx.halo >= 0) &&
                        q >= cx.cb - (SF_T - sf_stage<S>::depth) && q < cx.ce + (SF_T - sf_stage<S>::depth);
  const bool row_ok = ((cx.jmask >> r) & 1u) != 0;
  // never under a branch: a plane this stage does not evaluate has zero records
  // (fields lacking dimensions are indexed by the dimensions they have, under a guard)
  return sf_stage<S>::load_aux_bio(cx.aux, (long long)(q + cx.halo), plane_ok, cx.ld_off[r], SF_PLANE_BYTES,
                                   cx.j0 + r, cx.k0, row_ok, cx.kvec_in);
}

#if SF_AUX_PASS
// SF_AUX_PASS (T = 2, both operators read the same auxiliary fields): stage 2 evaluates at step
// p + 1 the plane stage 1 evaluated at step p, so stage 1 leaves its auxiliary rows in LDS --
// slots private to the thread, [step parity][row][thread]: no barrier, no bank conflict -- and
// stage 2 takes them from there instead of requesting the fields a second time.
__device__ __forceinline__ void sf_aux_pass(const sf_ctx& cx, const int p, const int r,
                                            const typename sf_stage<1>::aux_row& ax) {
  sf_aux_passed v;
  __builtin_memcpy(&v, &ax, sizeof v);
  cx.aux_lds[((p & 1) * SF_RJ + r) * (SF_BX * SF_BY) + cx.ty * SF_BX + cx.tx] = v;
}
__device__ __forceinline__ sf_aux_passed sf_aux_take(const sf_ctx& cx, const int p, const int r) {
  return cx.aux_lds[(((p - 1) & 1) * SF_RJ + r) * (SF_BX * SF_BY) + cx.ty * SF_BX + cx.tx];
}
#endif

__device__ __forceinline__ void sf_load_plane(const sf_t* __restrict__ in, const sf_ctx& cx,
                                              const int p, sf_vec (&dst)[SF_RJ], const bool enabled = true) {
  (void)in;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) SF_LOAD_ROW_IF(enabled, dst[r], p, r);
}

// Stage 1 is done with row r of the input window's "prev" slot: the row takes its next plane.
template <int PH>
__device__ __forceinline__ void sf_refill_row(sf_state& st, const sf_ctx& cx, const int p, const int r,
                                              const bool load_next) {
  constexpr int iprev = PH % SF_SLOTS;
  (void)iprev;
#if SF_RING4
  // it receives row r of the plane after the ones in flight (the next plane is already in flight in the fourth
  // slot -- the next two in the fourth and fifth), which has two (three) steps to land -- and nothing is copied
  SF_LOAD_ROW_IF(load_next, st.w[0][iprev][r], p + 1 + SF_INFLIGHT, r);
#else
  // it receives row r of input plane p+1 -- loads are spread over stage 1 instead of issued in a burst
  SF_LOAD_ROW_IF(load_next, st.w[0][iprev][r], p + 1, r);
#endif
}

// What a stage reads of ONE source window at one step: the rows of the neighbouring thread rows (LDS), the
// edge columns of the neighbouring waves (LDS), and -- row by row -- the point's six neighbours.
template <int W, int PH>
struct sf_rowsrc {
  static constexpr int iprev = PH % SF_SLOTS, icur = (PH + 1) % SF_SLOTS, inext = (PH + 2) % SF_SLOTS;
  sf_vec jm, jpl;  // the row above the one being evaluated; the first row of the thread row below
#if SF_WPR > 1
  sf_t e_lo[SF_RJ], e_hi[SF_RJ];
#endif
  sf_vec c, im, ip, jp;  // of the current row
  sf_t km_e, kp_e;       // k-1 of the vector's first element, k+1 of its last

  __device__ __forceinline__ void begin(const sf_state& st, const sf_t* lds, const sf_ctx& cx) {
    const int tx = cx.tx, ty = cx.ty;
    // first / last row of the neighbouring thread rows (LDS)
    jm = st.w[W][icur][0];
    jpl = st.w[W][icur][SF_RJ - 1];
    if constexpr (!SF_NOJ) {
      // No test of the thread row: the first / last thread row of the tile reads an image that
      // exists (its own) -- rows 0 and SF_RJ-1 there are halo rows, whatever they take as their
      // outer neighbour never reaches a stored value.  A divergent `if` here was the construct on
      // which the toolchain fault of DESIGN.md 5.1 showed.
      jm = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(W, ty > 0 ? ty - 1 : 0, 1) + tx * SF_VK]);
      jpl = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(W, ty < SF_BY - 1 ? ty + 1 : ty, 0) + tx * SF_VK]);
    }
#if SF_WPR > 1
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r) {
      e_lo[r] = lds[sf_edge_at(W, ty, r, cx.wave - 1, 1)];
      e_hi[r] = lds[sf_edge_at(W, ty, r, cx.wave + 1, 0)];
    }
#endif
  }

  // row r becomes the current row (the row above it is `jm`, left there by `next`)
  __device__ __forceinline__ void row(const sf_state& st, const sf_t* lds, const sf_ctx& cx, const int r) {
    (void)lds;
    (void)cx;
    c = st.w[W][icur][r];
    im = st.w[W][iprev][r];
    ip = st.w[W][inext][r];
    jp = (r < SF_RJ - 1) ? st.w[W][icur][r < SF_RJ - 1 ? r + 1 : r] : jpl;
    // innermost-dimension halo: adjacent lanes hold the adjacent vectors
#if SF_WPR > 1
    // Lanes 0 / 63 take the neighbouring wave's edge column -- or, from the
    // virtual waves beside the row, the boundary constant -- as the DPP move's
    // starting destination: no test, no select (the words were read before the
    // first row, see e_lo / e_hi).
    km_e = sf_neighbour_lane_or<true>(c[SF_VK - 1], e_lo[r]);
    kp_e = sf_neighbour_lane_or<false>(c[0], e_hi[r]);
#else
    // One wave per row: lanes 0 / 63 have no source lane.  Their value is the boundary constant the window's
    // readers declare; it is handed to the DPP move as the starting destination, so no select follows -- and a
    // boundary constant of +0 is what bound_ctrl writes.
    (void)lds;
    if (sf_win<W>::bc_zero) {
      km_e = sf_neighbour_lane<true>(c[SF_VK - 1]);
      kp_e = sf_neighbour_lane<false>(c[0]);
    } else {
      km_e = sf_neighbour_lane_or<true>(c[SF_VK - 1], (sf_t)sf_win<W>::bc());
      kp_e = sf_neighbour_lane_or<false>(c[0], (sf_t)sf_win<W>::bc());
    }
#endif
  }
  __device__ __forceinline__ sf_t km(const int v) const { return (v > 0) ? c[v > 0 ? v - 1 : 0] : km_e; }
  __device__ __forceinline__ sf_t kp(const int v) const { return (v < SF_VK - 1) ? c[v < SF_VK - 1 ? v + 1 : v] : kp_e; }
  // a row that is not evaluated still is the next row's j-neighbour
  __device__ __forceinline__ void skip(const sf_state& st, const int r) { jm = st.w[W][icur][r]; }
  __device__ __forceinline__ void next() { jm = c; }
};

// One stage of the fused group at one step.  A group is a DAG of SF_NS stages over SF_NW register windows
// (window 0: the input field; window sf_stage<S>::dst: what stage S hands to later stages of the group):
// stage S reads window sf_stage<S>::src (and, a join, sf_stage<S>::src2) at phase PH and produces plane
// q = p - depth of its field -- into its window, to HBM (output sf_stage<S>::out), or both (an intermediate
// the group hands on AND materialises).  A chain is the special case src = S - 1, dst = S, depth = S.
template <int S, int PH>
__device__ __forceinline__ void sf_stage_step(sf_state& st, const sf_t* lds, const sf_scalars& sc,
                                              const sf_outptrs& outs, const sf_ctx& cx, const int p,
                                              const bool load_next = false) {
  using stage = sf_stage<S>;
  constexpr int src = stage::src, src2 = stage::src2, dst = stage::dst, depth = stage::depth;
  constexpr bool joins = src2 >= 0;
  constexpr int iprev = PH % SF_SLOTS, icur = (PH + 1) % SF_SLOTS, inext = (PH + 2) % SF_SLOTS;
  (void)icur;
  sf_rowsrc<src, PH> A;
  A.begin(st, lds, cx);
  sf_rowsrc<(joins ? src2 : src), PH> B;
  if constexpr (joins) B.begin(st, lds, cx);
  // plane this stage produces (local owned coords)
  const int q = p - depth;
  const bool plane_in = (q + cx.goff >= 0) && (q + cx.goff < SF_N0G);
  const bool store_plane = (stage::out >= 0) && q >= cx.cb && q < cx.ce && plane_in;
  sf_t pad = (sf_t)0;
  if constexpr (dst >= 0) pad = sf_win<(dst >= 0 ? dst : 0)>::bc();
#if SF_AUX_AHEAD
  // all auxiliary rows of this stage are requested before its first row is
  // evaluated (they come straight from HBM: issued late they stall every row)
  typename sf_stage<S>::aux_row axs[SF_RJ];
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
#if SF_AUX_AHEAD == 2
    // requested during the previous step; the slot then takes the next plane's row
    axs[r] = static_cast<sf_auxslots<S>&>(st).a[r];
#if SF_AUX_CARRY
    // ... which is the row the previous stage has just used (it evaluated this step the plane
    // this stage evaluates in the next one): handed on in registers, not requested again
    if constexpr (sf_stage<S>::aux_carry_in)
      __builtin_memcpy(&static_cast<sf_auxslots<S>&>(st).a[r], &static_cast<sf_auxslots<S - 1>&>(st).used[r],
                       sizeof(typename sf_stage<S>::aux_row));
    else
#endif
    static_cast<sf_auxslots<S>&>(st).a[r] = sf_aux_row<S>(cx, q + 1, r);
#if SF_AUX_CARRY
    if constexpr (sf_stage<S>::aux_carry_out) static_cast<sf_auxslots<S>&>(st).used[r] = axs[r];
#endif
#else
#if SF_AUX_PASS
    if constexpr (sf_stage<S>::aux_from_prev) axs[r] = sf_aux_take(cx, p, r);
    else
#endif
    axs[r] = sf_aux_row<S>(cx, q, r);
#endif
  }
#endif
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    A.row(st, lds, cx, r);
    if constexpr (joins) B.row(st, lds, cx, r);
    // centre-only auxiliary fields of this stage, row r of plane q
#if SF_AUX_AHEAD
    const auto ax = axs[r];
#if SF_AUX_PASS
    if constexpr (sf_stage<S>::aux_to_next) sf_aux_pass(cx, p, r, ax);
#endif
#else
    const auto ax = sf_aux_row<S>(cx, q, r);
#endif
    // `copy` boundaries: which neighbours of this row's points lie outside the domain (the functor then
    // takes the centre value in their place; what the windows hold there is never used)
    unsigned edge_row = 0;
    if constexpr (sf_stage<S>::bc_copy)
      edge_row = (q + cx.goff <= 0 ? 1u : 0u) | (q + cx.goff >= SF_N0G - 1 ? 2u : 0u) |
                 (!SF_NOJ && cx.j0 + r <= 0 ? 4u : 0u) | (!SF_NOJ && cx.j0 + r >= SF_N1 - 1 ? 8u : 0u);
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) {
      unsigned edge = 0;
      if constexpr (sf_stage<S>::bc_copy)
        edge = edge_row | (cx.k0 + v <= 0 ? 16u : 0u) | (cx.k0 + v >= SF_N2 - 1 ? 32u : 0u);
      if constexpr (joins)
        o[v] = stage::apply2(A.c[v], A.im[v], A.ip[v], A.jm[v], A.jp[v], A.km(v), A.kp(v), B.c[v], B.im[v], B.ip[v],
                             B.jm[v], B.jp[v], B.km(v), B.kp(v), sc, ax, v, edge);
      else
        o[v] = stage::apply(A.c[v], A.im[v], A.ip[v], A.jm[v], A.jp[v], A.km(v), A.kp(v), sc, ax, v, edge);
    }
    A.next();
    if constexpr (joins) B.next();
    if constexpr (stage::refill) sf_refill_row<PH>(st, cx, p, r, load_next);
    if constexpr (stage::out >= 0) {
      // a materialised field: write interior, in-domain points
      sf_t* __restrict__ out = outs.p[stage::out >= 0 ? stage::out : 0];
      // always issued: a plane that is not stored has a resource of zero records,
      // rows and lanes that are not stored an offset outside the plane
      char* base = reinterpret_cast<char*>(out) + (long long)(q + cx.halo) * (long long)SF_PLANE_BYTES;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          base, 0, store_plane ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
      sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
    }
    if constexpr (dst >= 0) {
      // pad: outside the global domain the readers of this window must read THEIR constant
      // (skipped by a block-uniform branch for tiles and planes strictly inside)
      if (!(cx.tile_inside && plane_in)) {
        const bool row_in = plane_in && ((cx.jmask >> r) & 1u);
#pragma unroll
        for (int v = 0; v < SF_VK; ++v) o[v] = (row_in && ((cx.kmask >> v) & 1u)) ? o[v] : pad;
      }
      // becomes plane "next" of the window
      st.w[dst >= 0 ? dst : 0][inext][r] = o;
    }
#if SF_ROW_FENCE
    __builtin_amdgcn_sched_barrier(0);  // rows in order: bounds the live f64 temporaries
#endif
  }
}

// stages S, S + 1, ..., SF_NS in order (every stage after the ones it reads); the stage that is the last reader of
// the input window also moves that window on (codegen: sf_stage<S>::refill -- stage 1 of a chain)
template <int S, int PH>
__device__ __forceinline__ void sf_stages_from(sf_state& st, const sf_t* lds, const sf_t* __restrict__ in,
                                               const sf_scalars& sc, const sf_outptrs& outs, const sf_ctx& cx,
                                               const int p, const int p_end, const bool load_next) {
  if constexpr (S <= SF_NS) {
    sf_stage_step<S, PH>(st, lds, sc, outs, cx, p, sf_stage<S>::refill && load_next);
    if constexpr (sf_stage<S>::refill) {
    }
    (void)in;
    (void)p_end;
    sf_stages_from<S + 1, PH>(st, lds, in, sc, outs, cx, p, p_end, load_next);
  }
}

// stages S, S-1, ..., 2 (2-D chains: the storing stage first)
template <int S, int PH>
__device__ __forceinline__ void sf_later_stages_desc(sf_state& st, const sf_t* lds,
                                                     const sf_scalars& sc, const sf_outptrs& outs,
                                                     const sf_ctx& cx, const int p) {
  if constexpr (S >= 2) {
    sf_stage_step<S, PH>(st, lds, sc, outs, cx, p);
    sf_later_stages_desc<S - 1, PH>(st, lds, sc, outs, cx, p);
  }
}

#if SF_WPR > 1
template <int W>
__device__ __forceinline__ void sf_edge_prefill(sf_t* lds_all, const sf_ctx& cx) {
  if constexpr (W < SF_NW) {
    if (cx.lane == 0 && (cx.wave == 0 || cx.wave == SF_WPR - 1)) {
#pragma unroll
      for (int image = 0; image < (1 ? 2 : 1); ++image)
#pragma unroll
        for (int r = 0; r < SF_RJ; ++r) {
          sf_t* lds = lds_all + image * SF_IMAGE_ELEMS;
          if (cx.wave == 0) lds[sf_edge_at(W, cx.ty, r, -1, 1)] = sf_win<W>::bc();
          if (cx.wave == SF_WPR - 1) lds[sf_edge_at(W, cx.ty, r, SF_WPR, 0)] = sf_win<W>::bc();
        }
    }
    sf_edge_prefill<W + 1>(lds_all, cx);
  }
}
#endif

#if SF_AUX_AHEAD == 2
// Fill every stage's auxiliary slot with the rows its first step uses.
template <int S>
__device__ __forceinline__ void sf_aux_preload(sf_state& st, const sf_ctx& cx, const int p_first) {
  if constexpr (S <= SF_NS) {
    const int q = p_first - sf_stage<S>::depth;
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r) static_cast<sf_auxslots<S>&>(st).a[r] = sf_aux_row<S>(cx, q, r);
    sf_aux_preload<S + 1>(st, cx, p_first);
  }
}
#endif


// One step (input plane p) at phase PH.

template <int PH>
__device__ __forceinline__ void sf_step(sf_state& st, sf_t* lds, const sf_t* __restrict__ in,
                                        const sf_outptrs& outs, const sf_scalars& sc,
                                        const sf_ctx& cx, const int p, const int p_end) {
  constexpr int icur = (PH + 1) % SF_SLOTS;
  // Make the window opaque at the step boundary: otherwise the compiler keeps
  // the f64 conversions of whole planes alive from one unrolled step to the
  // next (fewer v_cvt, but ~80 more VGPRs and spills).
#if SF_OPAQUE
#pragma unroll
  for (int s = 0; s < SF_NW; ++s)
#pragma unroll
    for (int w = 0; w < SF_SLOTS; ++w)
#pragma unroll
      for (int r = 0; r < SF_RJ; ++r) {
        // (SF_RING4: nor the fourth slot -- in flight for the input, dead otherwise)
        if (SF_RING4 && w == (PH + 3) % SF_SLOTS) continue;
        // (a one-element vector is not a register operand: name its element)
        if constexpr (SF_VK == 1) asm volatile("" : "+v"(st.w[s][w][r][0]));
        else asm volatile("" : "+v"(st.w[s][w][r]));
      }
#endif
  // publish the rows / columns other threads need of every window's current plane
#pragma unroll
  for (int s = 0; s < SF_NW; ++s) {
    if constexpr (!SF_NOJ) {
      *reinterpret_cast<sf_vec*>(&lds[sf_rows_at(s, cx.ty, 0) + cx.tx * SF_VK]) = st.w[s][icur][0];
      *reinterpret_cast<sf_vec*>(&lds[sf_rows_at(s, cx.ty, 1) + cx.tx * SF_VK]) =
          st.w[s][icur][SF_RJ - 1];
    }
    if (SF_WPR > 1) {
      if (cx.lane == 0) {
#pragma unroll
        for (int r = 0; r < SF_RJ; ++r) lds[sf_edge_at(s, cx.ty, r, cx.wave, 0)] = st.w[s][icur][r][0];
      }
      if (cx.lane == 63) {
#pragma unroll
        for (int r = 0; r < SF_RJ; ++r)
          lds[sf_edge_at(s, cx.ty, r, cx.wave, 1)] = st.w[s][icur][r][SF_VK - 1];
      }
    }
  }
  if (SF_USE_LDS) __syncthreads();
  // the stages in order.  The last reader of the input window (stage 1 of a chain) consumes input plane p (slot
  // "next") and frees slot "prev", which receives the next plane the ring / staging / plain scheme asks for
#if SF_RING4
  const bool load_next = p + 1 + SF_INFLIGHT < p_end;
#else
  const bool load_next = p + 1 < p_end;
#endif
  sf_stages_from<1, PH>(st, lds, in, sc, outs, cx, p, p_end, load_next);
  if (SF_USE_LDS && !1) __syncthreads();
}


extern "C" __global__ void __launch_bounds__(SF_BX* SF_BY)
    SF_KERNEL_NAME(const sf_t* __restrict__ in, sf_t* __restrict__ out, sf_scalars sc,
                   sf_auxptrs aux, int halo,
                   int goff, int i_begin, int i_end, int li, int nch1, int i_begin2, int i_end2
#if SF_NOUT > 1
                   ,
                   sf_more_outs more
#endif
    ) {
  static_assert(!(SF_DAG && 0), "DAG groups run the stages in order");
  sf_outptrs outs;
  outs.p[0] = out;
#if SF_NOUT > 1
#pragma unroll
  for (int i = 1; i < SF_NOUT; ++i) outs.p[i] = static_cast<sf_t*>(more.p[i - 1]);
#endif
  // two exchange images used alternately -> one barrier per step
  __shared__ sf_t lds_all[2 * SF_IMAGE_ELEMS];
#if SF_AUX_PASS
  __shared__ sf_aux_passed lds_aux[2 * SF_RJ * SF_BX * SF_BY];
#endif

  sf_ctx cx;
  cx.in = in;
  cx.aux = aux;
#if SF_AUX_PASS
  cx.aux_lds = lds_aux;
#endif
  const int sf_tid_x = (int)threadIdx.x, sf_tid_y = (int)threadIdx.y;
  cx.tx = sf_tid_x;
  // thread row and wave-within-row are the same for all lanes of a wave (SF_BX is
  // a multiple of 64): telling the compiler so makes every test on them a scalar branch
  cx.ty = sf_tid_y;
  cx.wave = cx.tx >> 6;
  cx.lane = cx.tx & 63;
  cx.goff = goff;
  cx.halo = halo;

  // XCD-aware block order: consecutive logical tiles (adjacent in j, sharing
  // halo rows) land on one XCD and therefore one L2 (blocks are dealt
  // round-robin over the 8 XCDs; speed only, never correctness).
  const int nb = gridDim.x, b = blockIdx.x;
  const int xq = nb >> 3, xr = nb & 7, xcd = b & 7;
  const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  // k-tiles fastest: an XCD's share is a band of whole tile rows, every pair of k-neighbours in it
  const int kt = L % SF_NKT;
  const int jt = (L / SF_NKT) % SF_NJT;
  const int ch = L / (SF_NJT * SF_NKT);

  // chunks [0, nch1) cover planes [i_begin, i_end), later chunks a second range
  // [i_begin2, i_end2) (both slab boundaries in one launch)
  if (ch < nch1) {
    cx.cb = i_begin + ch * li;
    cx.ce = (cx.cb + li < i_end) ? cx.cb + li : i_end;
  } else {
    cx.cb = i_begin2 + (ch - nch1) * li;
    cx.ce = (cx.cb + li < i_end2) ? cx.cb + li : i_end2;
  }
  if (cx.cb >= cx.ce) return;

  cx.j0 = SF_NOJ ? 0 : (jt * SF_TJI - SF_T + cx.ty * SF_RJ);
  cx.k0 = SF_KTILED ? (kt * SF_TKI - SF_HK + cx.tx * SF_VK) : cx.tx * SF_VK;

  {
    const int tj0 = SF_NOJ ? 0 : (jt * SF_TJI - SF_T);
    const int tk0 = SF_KTILED ? (kt * SF_TKI - SF_HK) : 0;
    cx.tile_inside = tj0 >= 0 && tj0 + (SF_NOJ ? 1 : SF_TJH) <= SF_N1 && tk0 >= 0 && tk0 + SF_TKH <= SF_N2;
  }
  cx.jmask = 0;
  cx.kmask = 0;
  cx.store_mask = 0;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const int j = cx.j0 + r, tr = cx.ty * SF_RJ + r;
    const bool in_dom = (j >= 0) && (j < SF_N1);
    cx.jmask |= (in_dom ? 1u : 0u) << r;
    cx.store_mask |= ((in_dom && (SF_NOJ || (tr >= SF_T && tr < SF_TJH - SF_T))) ? 1u : 0u) << r;
  }
#pragma unroll
  for (int v = 0; v < SF_VK; ++v)
    cx.kmask |= ((cx.k0 + v >= 0 && cx.k0 + v < SF_N2) ? 1u : 0u) << v;
  cx.kvec_in = (cx.kmask & 1u) != 0;  // N2 % VK == 0: whole vector in or out
  if (SF_KTILED) {
    const int tk = cx.tx * SF_VK;
    if (!(tk >= SF_HK && tk < SF_TKH - SF_HK && cx.kvec_in)) cx.store_mask = 0;
  }

#if SF_NT & 4
  {
    unsigned excl = 0;
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r) {
      const int tr = sf_tid_y * SF_RJ + r;
      excl |= ((!SF_KTILED && !SF_NOJ && tr >= 2 * SF_T && tr < SF_TJH - 2 * SF_T) ? 1u : 0u) << r;
    }
    cx.nt_rows = (unsigned)__builtin_amdgcn_readfirstlane((int)excl);
  }
#endif
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const unsigned off = (unsigned)(((cx.j0 + r) * SF_N2 + cx.k0) * (int)sizeof(sf_t));
    cx.ld_off[r] = (((cx.jmask >> r) & 1u) && cx.kvec_in) ? off : SF_OOB;
    cx.st_off[r] = ((cx.store_mask >> r) & 1u) ? off : SF_OOB;
  }

  sf_state st;
#pragma unroll
  for (int s = 0; s < SF_NW; ++s)
#pragma unroll
    for (int w = 0; w < SF_SLOTS; ++w)
#pragma unroll
      for (int r = 0; r < SF_RJ; ++r) st.w[s][w][r] = (sf_vec)(sf_t)0;

  // p_end bounds the input planes read and the steps
  const int p_begin = cx.cb - SF_T, p_end = cx.ce + SF_T;
  const int p_last = p_end;
  sf_load_plane(in, cx, p_begin, st.w[0][2]);  // slot "next" of phase 0
#if SF_RING4
  // the plane(s) after the one stage 1 starts with are already in flight (fourth / fifth slot)
  sf_load_plane(in, cx, p_begin + 1, st.w[0][3], p_begin + 1 < p_end);
#endif

#if SF_AUX_AHEAD == 2
  sf_aux_preload<1>(st, cx, p_begin);
#endif
#if SF_WPR > 1
  // the virtual waves' edge words: window s is read by stage s + 1, whose boundary
  // constant they hold (never overwritten; the first step's barrier orders them)
  sf_edge_prefill<0>(lds_all, cx);
#endif
  // Two exchange images alternate every step (run-time offset); the window
  // phase cycles with period 3 (compile-time slot indices).
  int image = 0;
  // the trip always runs three steps: up to two surplus steps past p_end
  // compute planes nobody stores (loads and stores are range-guarded), which
  // keeps the loop body free of control flow between the phases
  // (SF_RING4: four steps, up to three surplus ones)
  for (int p = p_begin; p < p_last; p += SF_SLOTS) {
    sf_step<0>(st, lds_all + image, in, outs, sc, cx, p, p_end);
    image = SF_IMAGE_ELEMS - image;
    sf_step<1>(st, lds_all + image, in, outs, sc, cx, p + 1, p_end);
    image = SF_IMAGE_ELEMS - image;
    sf_step<2>(st, lds_all + image, in, outs, sc, cx, p + 2, p_end);
    image = SF_IMAGE_ELEMS - image;
#if SF_RING4
    sf_step<3>(st, lds_all + image, in, outs, sc, cx, p + 3, p_end);
    image = SF_IMAGE_ELEMS - image;
#endif
  }
}
