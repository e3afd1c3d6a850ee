// compact3d.h — fused T-stage plane-streaming kernel for COMPACT stencils on a
// 3-D field (CDNA4 / gfx950, wave64): every operator reads its streamed source
// through any subset of the 27 offsets {-1,0,1}^3 (the `box` shape of the
// reference's workload generator, bin/synthesize.py:19-31,103-104) and may read ONE
// further full field -- a program input -- through such offsets as well (the
// generator's extra spatial fields, bin/synthesize.py:170-196).  Radius-1 *star*
// chains over a single field stay on kernels/star3d.h, which this kernel extends.
// Compiled at plan creation by hipRTC with the macros and `sf_stage<S>` functors
// emitted by codegen (codegen.hpp: gen_compact).
//
// Per-point semantics are those of ExpandStencilCPU (reference
// stencilflow/stencil/cpu.py:58-115): an access whose offset leaves the global
// domain in ANY dimension yields the operator's boundary constant for that field.
// Implemented as padding by coordinate, exactly as in star3d.h: every value a
// stage hands on (and every plane loaded) is replaced by the consumer's constant
// at coordinates outside the global domain, so corner accesses need no case of
// their own.
//
// Decomposition (as star3d.h)
//   block  = tile of TJH x TKH points of the (j,k) plane incl. SF_T halo rows (and
//            SF_HK halo columns when k-tiled), marching along i over one chunk;
//   thread = SF_RJ consecutive rows x SF_VK consecutive k;
//   window = for every stage boundary (and every extra field) FOUR register slots of
//            one plane each, rotating with period 4 (the step loop is unrolled by 4,
//            slot indices are compile-time): prev, cur, next and the plane being
//            produced / in flight.
//
// Step order.  A diagonal neighbour (i+-1 together with j+-1 or k+-1) lives in
// ANOTHER thread's registers of the prev / next plane, so a stage can only
// consume planes whose edge rows and columns have been published to LDS -- i.e.
// planes finished in an EARLIER step.  At step p stage S therefore produces plane
// p - (2S - 1) (star3d.h's "storing stage first" order): all stages of a step are
// independent of each other, and ONE barrier per step separates publishing from
// consuming.
//
// LDS images.  Per window a ring of images, one written per step:
//   diagonal windows (the consumer reads prev/next planes off its own (j,k)):
//            the plane that has just become `next` is published, ring of 4
//            (three are being read, one is written);
//   star-like windows: the `cur` plane is published, ring of 2 (star3d.h's double buffer).
// An image holds the first and last row of every thread row and, per row, the
// first / last element of every wave (+ one virtual wave on either side holding
// the consumer's boundary constant, so no wave tests whether its neighbour exists).
//
// Memory instructions are branch-free buffer loads / stores over a resource that
// describes exactly one plane (see star3d.h, SF_BUFFER_IO): lanes, rows and whole
// planes that must not touch memory get an offset outside the resource.
//
// 2-D programs (SF_NOJ: stream axis = rows, one row per thread, SF_BY = SF_RJ = 1,
// k-tiled strips): offsets are {-1,0,1}^2 -- the 9-point `box` of the generator in
// 2-D -- only the wave-edge words of an image are used.
//
// Macros from codegen: SF_T SF_NX SF_VK SF_RJ SF_BX SF_BY SF_HK SF_KTILED SF_NOJ SF_N0G
//   SF_N1 SF_N2 SF_NJT SF_NKT SF_NT SF_OPAQUE SF_ROW_FENCE SF_KERNEL_NAME;
//   typedef sf_t, struct sf_scalars, struct sf_auxptrs (extra-field pointers),
//   template<int S> struct sf_stage {need, xneed, xwin, xarg, bc(), xbc(), apply()}.

typedef sf_t sf_vec __attribute__((ext_vector_type(SF_VK)));
typedef unsigned sf_u4 __attribute__((ext_vector_type(4)));
typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));

#define SF_OOB 0x80000000u
#define SF_PLANE_ELEMS ((long long)SF_N1 * (long long)SF_N2)
#define SF_PLANE_BYTES ((unsigned)(SF_PLANE_ELEMS * (long long)sizeof(sf_t)))
#define SF_RSRC_FLAGS 0x00020000 /* raw buffer, 32-bit data format (gfx9 / CDNA) */

#define SF_SLOTS 4
#define SF_NWIN (SF_T + SF_NX)
#define SF_TJH (SF_BY * SF_RJ)
#define SF_TKH (SF_BX * SF_VK)
// Lane neighbours (k-1 of a thread's first element, k+1 of its last): DPP wave_shr / wave_shl with the wave-edge words
// from LDS.  (Round 4 built two alternatives -- no cross-lane vector instruction at all, the rows going through
// ds_swizzle / shifted reads of the row images; all the moves of a stage step in one burst -- bit-exact and slower on
// the 27-point box at every tile shape, profiles/r04_box_xlane.log; removed in round 5, NOTES.md.)
#define SF_WPR (SF_BX / 64) /* waves per row */
#define SF_ROW_STRIDE SF_TKH
#if SF_NOJ
#define SF_TJI 1  // 2-D programs: the stream axis is j, there is no tiled row axis (one row per thread)
#else
#define SF_TJI (SF_TJH - 2 * SF_T)
#endif
#define SF_TKI (SF_TKH - 2 * SF_HK)

// One image of one window = the first and last row of every thread row
// (SF_ROWS_ELEMS) + the wave-edge words of every row, for SF_BY + 2 thread rows and
// SF_WPR + 2 waves per row: the virtual thread rows -1 / SF_BY are never written
// (what is read there only reaches halo rows, whose results are discarded) and the
// virtual waves -1 / SF_WPR hold the consumer's boundary constant -- so no thread
// tests whether a neighbour exists, and every address is ONE run-time base per
// kind plus a compile-time offset.  The edge words of all images come first (a few
// KB: one base register reaches all of them through the 16-bit offset field),
// the row images follow.
#define SF_ROWS_ELEMS (SF_BY * 2 * SF_ROW_STRIDE)
#define SF_EDGE_WAVES (SF_WPR + 2)
#define SF_EDGE_ELEMS ((SF_BY + 2) * SF_RJ * SF_EDGE_WAVES * 2)
#define SF_WIN_ELEMS (SF_ROWS_ELEMS + SF_EDGE_ELEMS)

template <int I>
struct sf_ic {
  static constexpr int value = I;
};
template <int B, int E, typename F>
__device__ __forceinline__ void sf_static_for(F&& f) {
  if constexpr (B < E) {
    f(sf_ic<B>{});
    sf_static_for<B + 1, E>(f);
  }
}

// bit of offset (d, e, f) = (di + 1, dj + 1, dk + 1) in a stage's `need` mask
__host__ __device__ constexpr unsigned sf_bit(int d, int e, int f) { return 1u << ((d * 3 + e) * 3 + f); }
// does the mask touch plane d at all / off the thread's own (j,k) / through a lane neighbour?
__host__ __device__ constexpr bool sf_needs_plane(unsigned m, int d) { return ((m >> (d * 9)) & 0x1ffu) != 0; }
__host__ __device__ constexpr bool sf_needs_row(unsigned m, int d, int e) { return ((m >> ((d * 3 + e) * 3)) & 7u) != 0; }
__host__ __device__ constexpr bool sf_needs_lateral(unsigned m, int d) {
  return (((m >> (d * 9)) & 0x1ffu) & ~(1u << 4)) != 0;  // anything but (e, f) = (1, 1)
}
// diagonal window: the consumer reads the prev or next plane off its own (j,k)
__host__ __device__ constexpr bool sf_diag(unsigned m) { return sf_needs_lateral(m, 0) || sf_needs_lateral(m, 2); }

// ---- which stage consumes window W, and through which mask -------------------
// windows 0..T-1: data of stage W (0 = the input field), consumed by stage W + 1;
// windows T..T+NX-1: extra fields, consumed by the stage whose xwin names them.
template <int W, int S = 1>
struct sf_win_info {
  static constexpr bool mine = (W < SF_T) ? (S == W + 1) : (sf_stage<S>::xwin == W);
  using next = sf_win_info<W, (S < SF_T ? S + 1 : S)>;
  static constexpr unsigned mask =
      mine ? ((W < SF_T) ? sf_stage<S>::need : sf_stage<S>::xneed) : ((S < SF_T) ? sf_win_info<W, S + 1>::mask : 0u);
  static constexpr int consumer = mine ? S : ((S < SF_T) ? sf_win_info<W, S + 1>::consumer : 0);
};
template <int W>
struct sf_win_info<W, SF_T + 1> {
  static constexpr unsigned mask = 0u;
  static constexpr int consumer = 0;
};

template <int W>
struct sf_win {
  static constexpr unsigned mask = sf_win_info<W>::mask;
  static constexpr bool diag = sf_diag(mask);
  static constexpr bool lateral = sf_needs_lateral(mask, 0) || sf_needs_lateral(mask, 1) || sf_needs_lateral(mask, 2);
  static constexpr int ring = diag ? 4 : 2;  // images
};

// number of the first image of window W's ring among all images
template <int W>
struct sf_win_base {
  static constexpr int value = sf_win_base<W - 1>::value + (sf_win<W - 1>::lateral ? sf_win<W - 1>::ring : 0);
};
template <>
struct sf_win_base<0> {
  static constexpr int value = 0;
};
#define SF_IMAGES (sf_win_base<SF_NWIN>::value)
#define SF_LDS_ELEMS (SF_IMAGES > 0 ? SF_IMAGES * SF_WIN_ELEMS : 1)
// element offsets of image `g` (numbered over all windows)
#define SF_EDGE_IMAGE(g) ((g) * SF_EDGE_ELEMS)
#define SF_ROWS_IMAGE(g) (SF_IMAGES * SF_EDGE_ELEMS + (g) * SF_ROWS_ELEMS)

struct sf_state {
  sf_vec w[SF_NWIN][SF_SLOTS][SF_RJ];
};

struct sf_ctx {
  const sf_t* in;
  sf_auxptrs xp;
  int tx, ty, lane, wave;  // lane / wave: within / index of the SF_SEG-lane segment of the row
  bool seg_first, seg_last;
  // run-time parts of the LDS addresses (elements): this thread's vector in the first
  // / last row of the thread row below / above (clamped at the tile's ends), in its
  // own rows, and the edge word (thread row ty - 1, row 0, wave - 1, side 0)
  int row_lo, row_hi, row_own, edge0;
  unsigned jmask, kmask;
  bool kvec_in, tile_inside;
  int goff, halo, cb, ce, j0, k0;
  unsigned ld_off[SF_RJ], st_off[SF_RJ];
};

__host__ __device__ constexpr int sf_rows_at(int ty, int which) { return (ty * 2 + which) * SF_ROW_STRIDE + 0; }
// (ty in -1 .. SF_BY, w in -1 .. SF_WPR)
__host__ __device__ constexpr int sf_edge_at(int ty, int r, int w, int side) {
  return ((((ty + 1) * SF_RJ + r) * SF_EDGE_WAVES + (w + 1)) * 2 + side);
}

// value of the adjacent lane through the DPP data path; lanes without a source
// (lane 0 taking from below, lane 63 from above) keep `edge`
template <bool FROM_LOWER, typename T>
__device__ __forceinline__ T sf_neighbour_lane_or(T x, T edge) {
  constexpr int ctrl = FROM_LOWER ? 0x138 /* wave_shr:1 */ : 0x130 /* wave_shl:1 */;
  if constexpr (sizeof(T) == 4) {
    int moved = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x), ctrl, 0xf,
                                            0xf, false);
    asm("" : "+v"(moved));  // keeps LLVM's DPP combiner off it (star3d.h); not volatile: may move, may die
    return __builtin_bit_cast(T, moved);
  } else {
    const long long v = __builtin_bit_cast(long long, x), e = __builtin_bit_cast(long long, edge);
    int rlo = __builtin_amdgcn_update_dpp((int)(e & 0xffffffffll), (int)(v & 0xffffffffll), ctrl, 0xf, 0xf, false);
    int rhi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(v >> 32), ctrl, 0xf, 0xf, false);
    asm("" : "+v"(rlo), "+v"(rhi));
    return __builtin_bit_cast(T, ((long long)rhi << 32) | (unsigned int)rlo);
  }
}

// value of the adjacent lane within its 32-lane segment through the LDS crossbar (no LDS memory is touched);
// lanes at the segment's end receive the value of its other end and replace it (sf_merge_row)
template <bool FROM_LOWER, typename T>
__device__ __forceinline__ T sf_rotate_lane(T x) {
  constexpr int pattern = FROM_LOWER ? 0xC420 /* swizzle(ROTATE, 1, 1): lane i reads i - 1 */ : 0xC020 /* reads i + 1 */;
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), pattern));
  } else {
    const long long v = __builtin_bit_cast(long long, x);
    const int rlo = __builtin_amdgcn_ds_swizzle((int)(v & 0xffffffffll), pattern);
    const int rhi = __builtin_amdgcn_ds_swizzle((int)(v >> 32), pattern);
    return __builtin_bit_cast(T, ((long long)rhi << 32) | (unsigned int)rlo);
  }
}

template <typename V, int aux>
__device__ __forceinline__ V sf_buf_load(const __amdgpu_buffer_rsrc_t rs, const unsigned off) {
  if constexpr (sizeof(V) == 4) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, aux));
  } else if constexpr (sizeof(V) == 8) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, aux));
  } else if constexpr (sizeof(V) == 16) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, aux));
  } else {
    static_assert(sizeof(V) == 32, "vector of 4, 8, 16 or 32 bytes");
    struct {
      sf_u4 lo, hi;
    } two;
    two.lo = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, aux);
    two.hi = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, aux);
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
    static_assert(sizeof(V) == 32, "vector of 4, 8, 16 or 32 bytes");
    struct Two {
      sf_u4 lo, hi;
    };
    const Two two = __builtin_bit_cast(Two, v);
    __builtin_amdgcn_raw_buffer_store_b128(two.lo, rs, off, 0, aux);
    __builtin_amdgcn_raw_buffer_store_b128(two.hi, rs, off + 16u, 0, aux);
  }
}

// Row r of plane p of `field` (local plane coordinates), padded with `pad` outside
// the global domain; `enabled` false: nothing is read (the row holds the padding).
template <bool PAD_ZERO>
__device__ __forceinline__ sf_vec sf_load_row(const sf_ctx& cx, const sf_t* field, const int p, const int r,
                                              const bool enabled, const sf_t pad) {
  const bool plane_ok = enabled && (p + cx.goff >= 0) && (p + cx.goff < SF_N0G);
  const char* base = reinterpret_cast<const char*>(field) + (long long)(p + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, plane_ok ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
  sf_vec v = sf_buf_load<sf_vec, 0>(rs, cx.ld_off[r]);
  if constexpr (!PAD_ZERO) {
    const bool ok = plane_ok && cx.ld_off[r] != SF_OOB;
#pragma unroll
    for (int e = 0; e < SF_VK; ++e) v[e] = ok ? v[e] : pad;
  }
  return v;
}

// Per stage and step: what a thread needs of one window beyond its own registers --
// rows -1 and SF_RJ of the three planes (the neighbouring thread rows', from LDS)
// and, for every source row -1..SF_RJ, the element left of its first and right of
// its last column (the adjacent lane's, through DPP; at a wave edge the
// neighbouring wave's, from LDS).  Filled row by row as the output rows advance
// (sf_prepare_row), so that every lane exchange is done once per source row, not
// once per output row that reads it.
struct sf_nbr {
  sf_vec lo[3], hi[3];
  sf_t km[3][SF_RJ + 2], kp[3][SF_RJ + 2];
};

__host__ __device__ constexpr unsigned sf_plane_bits(unsigned m, int d) { return (m >> (d * 9)) & 0x1ffu; }

// Source row RR (-1 .. SF_RJ) of window WIN at phase PH, in two halves so that the LDS latency is covered by the
// evaluation of a whole output row (SF_LDS_AHEAD; measured on the 27-point box: every second instruction the waves
// waited on was an `s_waitcnt lgkmcnt(0)` ten instructions behind its `ds_read2_b32`):
//   sf_fetch_row  issues the LDS reads -- the neighbouring thread row's row (RR = -1 / SF_RJ) and the two wave-edge
//                 words of the row, which land in the very registers that will hold k-1 / k+1;
//   sf_merge_row  exchanges the lane neighbours through DPP; the lanes at a wave edge keep the LDS word.
template <int WIN, unsigned NEED, int PH, int RR, int d>
struct sf_row_plan {
  static constexpr bool diag = sf_win<WIN>::diag;
  // image that holds plane d: diagonal windows publish `next` every step (ring of 4:
  // next is this step's image, cur the previous one, prev the one before); star-like
  // windows publish `cur` (ring of 2)
  static constexpr int image = d == 0 ? (diag ? (PH + 2) % 4 : 0) : (d == 1 ? (diag ? (PH + 3) % 4 : PH % 2) : (diag ? PH % 4 : 0));
  static constexpr bool below = RR < 0, above = RR >= SF_RJ;
  static constexpr unsigned bits = sf_plane_bits(NEED, d);
  // rows of this plane that some output row reads as its j-1 / j / j+1 neighbour
  static constexpr bool as_jm = (bits & 0x007u) != 0, as_jp = (bits & 0x1c0u) != 0;
  static constexpr bool want_row = below ? as_jm : (above ? as_jp : bits != 0);
  static constexpr bool want_km = want_row && (bits & 0x049u) != 0 && (!below || (bits & 0x001u)) && (!above || (bits & 0x040u));
  static constexpr bool want_kp = want_row && (bits & 0x124u) != 0 && (!below || (bits & 0x004u)) && (!above || (bits & 0x100u));
  static constexpr int g = sf_win_base<WIN>::value + image;
  // thread row that owns the row (relative to ty), and its index there
  static constexpr int dty = below ? -1 : (above ? 1 : 0);
  static constexpr int orow = below ? SF_RJ - 1 : (above ? 0 : RR);
  // cx.edge0 addresses (ty - 1, row 0, wave - 1, side 0): the rest is compile-time
  static constexpr int e_lo = sf_edge_at(dty, orow, -1, 1) - sf_edge_at(-1, 0, -1, 0);
  static constexpr int e_hi = sf_edge_at(dty, orow, 1, 0) - sf_edge_at(-1, 0, -1, 0);
};

template <int WIN, unsigned NEED, int PH, int RR>
__device__ __forceinline__ void sf_fetch_row(sf_nbr& nb, const sf_state& st, const sf_t* lds_all, const sf_ctx& cx) {
  if constexpr (RR >= -1 && RR <= SF_RJ) {
    sf_static_for<0, 3>([&](auto D) {
      using plan = sf_row_plan<WIN, NEED, PH, RR, decltype(D)::value>;
      constexpr int d = decltype(D)::value;
      if constexpr (plan::want_row) {
        if constexpr (plan::below) nb.lo[d] = *reinterpret_cast<const sf_vec*>(&lds_all[SF_ROWS_IMAGE(plan::g) + cx.row_lo]);
        if constexpr (plan::above) nb.hi[d] = *reinterpret_cast<const sf_vec*>(&lds_all[SF_ROWS_IMAGE(plan::g) + cx.row_hi]);
        if constexpr (plan::want_km) nb.km[d][RR + 1] = lds_all[SF_EDGE_IMAGE(plan::g) + cx.edge0 + plan::e_lo];
        if constexpr (plan::want_kp) nb.kp[d][RR + 1] = lds_all[SF_EDGE_IMAGE(plan::g) + cx.edge0 + plan::e_hi];
      }
    });
  }
}

template <int WIN, unsigned NEED, int PH, int RR>
__device__ __forceinline__ void sf_merge_row(sf_nbr& nb, const sf_state& st, const sf_ctx& cx) {
  constexpr int slot[3] = {PH % SF_SLOTS, (PH + 1) % SF_SLOTS, (PH + 2) % SF_SLOTS};
  sf_static_for<0, 3>([&](auto D) {
    using plan = sf_row_plan<WIN, NEED, PH, RR, decltype(D)::value>;
    constexpr int d = decltype(D)::value;
    if constexpr (plan::want_row) {
      sf_vec row;
      if constexpr (plan::below) row = nb.lo[d];
      else if constexpr (plan::above) row = nb.hi[d];
      else row = st.w[WIN][slot[d]][RR < 0 ? 0 : (RR >= SF_RJ ? SF_RJ - 1 : RR)];
      if constexpr (plan::want_km)  // k-1: the lane below, or the lower wave's last element
        nb.km[d][RR + 1] = sf_neighbour_lane_or<true>(row[SF_VK - 1], nb.km[d][RR + 1]);
      if constexpr (plan::want_kp)  // k+1
        nb.kp[d][RR + 1] = sf_neighbour_lane_or<false>(row[0], nb.kp[d][RR + 1]);
    }
  });
}

// LDS reads of the first source rows of a stage step (-1, 0, 1); issued by the stage BEFORE (at its last output row) or,
// for the first stage of a step, right behind the barrier.
template <int WIN, unsigned NEED, int PH>
__device__ __forceinline__ void sf_gather_begin(sf_nbr& nb, const sf_state& st, const sf_t* lds_all, const sf_ctx& cx) {
  sf_fetch_row<WIN, NEED, PH, -1>(nb, st, lds_all, cx);
  sf_fetch_row<WIN, NEED, PH, 0>(nb, st, lds_all, cx);
  sf_fetch_row<WIN, NEED, PH, 1>(nb, st, lds_all, cx);
}

// The source rows output row R needs that are not prepared yet: -1, 0 and 1 before the first output
// row, R + 1 afterwards (also run for an output row that is skipped: the rows after it build on it);
// the LDS reads of the row after that are issued for the next output row to find.
template <int WIN, unsigned NEED, int PH, int R>
__device__ __forceinline__ void sf_gather_prepare(sf_nbr& nb, const sf_state& st, const sf_t* lds_all, const sf_ctx& cx) {
  if constexpr (R == 0) {
    sf_merge_row<WIN, NEED, PH, -1>(nb, st, cx);
    sf_merge_row<WIN, NEED, PH, 0>(nb, st, cx);
  }
  sf_merge_row<WIN, NEED, PH, R + 1>(nb, st, cx);
  sf_fetch_row<WIN, NEED, PH, R + 2>(nb, st, lds_all, cx);
}

// Neighbourhood of output row R: n[v][(d*3+e)*3+f] for the SF_VK points of the row,
// read from window WIN through mask NEED at phase PH.
template <int WIN, unsigned NEED, int PH, int R>
__device__ __forceinline__ void sf_gather(const sf_nbr& nb, const sf_state& st, sf_t (&n)[SF_VK][27]) {
  constexpr int slot[3] = {PH % SF_SLOTS, (PH + 1) % SF_SLOTS, (PH + 2) % SF_SLOTS};
  sf_static_for<0, 3>([&](auto D) {
    constexpr int d = decltype(D)::value;
    sf_static_for<0, 3>([&](auto E) {
      constexpr int e = decltype(E)::value;
      constexpr unsigned bits = (NEED >> ((d * 3 + e) * 3)) & 7u;
      if constexpr (bits != 0) {
        constexpr int rr = R + e - 1;
        sf_vec row;
        if constexpr (rr < 0) row = nb.lo[d];
        else if constexpr (rr >= SF_RJ) row = nb.hi[d];
        else row = st.w[WIN][slot[d]][rr < 0 ? 0 : (rr >= SF_RJ ? SF_RJ - 1 : rr)];
#pragma unroll
        for (int v = 0; v < SF_VK; ++v) {
          if constexpr ((bits & 1u) != 0) n[v][(d * 3 + e) * 3 + 0] = (v > 0) ? row[v > 0 ? v - 1 : 0] : nb.km[d][rr + 1];
          if constexpr ((bits & 2u) != 0) n[v][(d * 3 + e) * 3 + 1] = row[v];
          if constexpr ((bits & 4u) != 0)
            n[v][(d * 3 + e) * 3 + 2] = (v < SF_VK - 1) ? row[v < SF_VK - 1 ? v + 1 : v] : nb.kp[d][rr + 1];
        }
      }
    });
  });
}

// Row R of the windows stage S loads from memory (the input window for stage 1, its
// extra field's window) is dead: request the plane after the one in flight.
template <int S, int PH, int R>
__device__ __forceinline__ void sf_refill(sf_state& st, const sf_ctx& cx, const int p, const int p_end) {
  using stage = sf_stage<S>;
  constexpr int iprev = PH % SF_SLOTS;
  if constexpr (S == 1) {
    // input window: prev (plane p-2) is dead, plane p+1 is in flight in the fourth
    // slot, so prev's row receives plane p+2 (two steps to land, no copy)
    st.w[0][iprev][R] = sf_load_row<stage::bc_zero>(cx, cx.in, p + 2, R, p + 2 < p_end, stage::bc());
  }
  if constexpr (stage::xneed != 0) {
    // extra field: this stage consumed planes q-1..q+1, q+2 is in flight: request q+3,
    // within what this chunk's stage S evaluates (+1 plane on either side)
    const int qx = p - (2 * S - 1) + 3;
    const bool want = qx >= cx.cb - (SF_T - S) - 1 && qx < cx.ce + (SF_T - S) + 1;
    st.w[stage::xneed != 0 ? stage::xwin : 0][iprev][R] = sf_load_row<stage::xbc_zero>(
        cx, static_cast<const sf_t*>(cx.xp.p[stage::xneed != 0 ? stage::xarg : 0]), qx, R, want, stage::xbc());
  }
}

// Stage S at step p (phase PH): produces plane q = p - (2S - 1) of stage-S data
// from window S-1 (planes q-1, q, q+1) and, where it has one, its extra field's
// window; the result goes into the free slot of window S (last stage: to HBM).
template <int S, int PH>
__device__ __forceinline__ void sf_stage_step(sf_state& st, const sf_t* lds_all, const sf_scalars& sc,
                                              sf_t* __restrict__ out, const sf_ctx& cx, const int p,
                                              const int p_end, sf_nbr& nbn, sf_nbr& nbx, sf_nbr& nbn_next,
                                              sf_nbr& nbx_next) {
  using stage = sf_stage<S>;
  constexpr int src = S - 1;
  constexpr int iprev = PH % SF_SLOTS, inew = (PH + 3) % SF_SLOTS;
  constexpr bool has_x = stage::xneed != 0;
  constexpr int xw = has_x ? stage::xwin : 0;
  const int q = p - (2 * S - 1);
  const bool plane_in = (q + cx.goff >= 0) && (q + cx.goff < SF_N0G);
  const bool store_plane = (S == SF_T) && q >= cx.cb && q < cx.ce && plane_in;
  sf_t pad = (sf_t)0;
  if constexpr (S < SF_T) pad = sf_stage<(S < SF_T ? S + 1 : S)>::bc();
  sf_static_for<0, SF_RJ>([&](auto RR) {
    constexpr int r = decltype(RR)::value;
    sf_gather_prepare<src, stage::need, PH, r>(nbn, st, lds_all, cx);
    if constexpr (has_x) sf_gather_prepare<xw, stage::xneed, PH, r>(nbx, st, lds_all, cx);
    if constexpr (r == SF_RJ - 1 && S > 1) {
      // the stage that runs next in this step (S - 1; the stages of a step are independent of each other)
      using below = sf_stage<(S > 1 ? S - 1 : 1)>;
      sf_gather_begin<(S > 1 ? S - 2 : 0), below::need, PH>(nbn_next, st, lds_all, cx);
      if constexpr (below::xneed != 0) sf_gather_begin<(below::xneed != 0 ? below::xwin : 0), below::xneed, PH>(nbx_next, st, lds_all, cx);
    }
    __builtin_amdgcn_sched_barrier(0);  // the reads above are issued before the row is evaluated, not behind it
    sf_t n[SF_VK][27], x[SF_VK][27];
#pragma unroll
    for (int v = 0; v < SF_VK; ++v)
#pragma unroll
      for (int b = 0; b < 27; ++b) {
        n[v][b] = (sf_t)0;
        x[v][b] = (sf_t)0;
      }
    sf_gather<src, stage::need, PH, r>(nbn, st, n);
    if constexpr (has_x) sf_gather<xw, stage::xneed, PH, r>(nbx, st, x);
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) o[v] = stage::apply(n[v], x[v], sc, q + cx.goff, cx.j0 + r, cx.k0 + v);
    // Rows of the planes that die with this step take the planes after next.  Output
    // row r + 1 still reads row r of the prev plane (its j-1 neighbour there), so
    // the row that is dead after output row r is r - 1; the last one follows the loop.
    if constexpr (r > 0) sf_refill<S, PH, r - 1>(st, cx, p, p_end);
    if constexpr (r == SF_RJ - 1) sf_refill<S, PH, r>(st, cx, p, p_end);
    if constexpr (S == SF_T) {
      char* base = reinterpret_cast<char*>(out) + (long long)(q + cx.halo) * (long long)SF_PLANE_BYTES;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(base, 0, store_plane ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
      sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
    } else {
      // outside the global domain the next stage must read ITS constant
      if (!(cx.tile_inside && plane_in)) {
        const bool row_in = plane_in && ((cx.jmask >> r) & 1u);
#pragma unroll
        for (int v = 0; v < SF_VK; ++v) o[v] = (row_in && ((cx.kmask >> v) & 1u)) ? o[v] : pad;
      }
      st.w[S < SF_T ? S : 0][inew][r] = o;  // `next` of window S from the coming step on
    }
#if SF_ROW_FENCE
    __builtin_amdgcn_sched_barrier(0);  // rows in order: bounds the live temporaries
#endif
  });
}

template <int S, int PH>
__device__ __forceinline__ void sf_stages_desc(sf_state& st, const sf_t* lds_all, const sf_scalars& sc,
                                               sf_t* __restrict__ out, const sf_ctx& cx, const int p, const int p_end,
                                               sf_nbr& nbn, sf_nbr& nbx) {
  if constexpr (S >= 1) {
    sf_nbr nbn_next, nbx_next;
    sf_stage_step<S, PH>(st, lds_all, sc, out, cx, p, p_end, nbn, nbx, nbn_next, nbx_next);
    sf_stages_desc<S - 1, PH>(st, lds_all, sc, out, cx, p, p_end, nbn_next, nbx_next);
  }
}

// publish, of window W, the plane its consumer will need from other threads
template <int W, int PH>
__device__ __forceinline__ void sf_publish(const sf_state& st, sf_t* lds_all, const sf_ctx& cx) {
  if constexpr (W < SF_NWIN) {
    if constexpr (sf_win<W>::lateral) {
      constexpr bool diag = sf_win<W>::diag;
      constexpr int slot = diag ? (PH + 2) % SF_SLOTS : (PH + 1) % SF_SLOTS;  // next : cur
      constexpr int g = sf_win_base<W>::value + (diag ? PH % 4 : PH % 2);
      if constexpr (!SF_NOJ) {
        *reinterpret_cast<sf_vec*>(&lds_all[SF_ROWS_IMAGE(g) + cx.row_own]) = st.w[W][slot][0];
        *reinterpret_cast<sf_vec*>(&lds_all[SF_ROWS_IMAGE(g) + cx.row_own + SF_ROW_STRIDE]) = st.w[W][slot][SF_RJ - 1];
      }
      // own edge words: (ty, r, wave, side) = edge0 + compile-time offset
      constexpr int own = sf_edge_at(0, 0, 0, 0) - sf_edge_at(-1, 0, -1, 0);
      if (cx.seg_first) {
#pragma unroll
        for (int r = 0; r < SF_RJ; ++r)
          lds_all[SF_EDGE_IMAGE(g) + cx.edge0 + own + r * SF_EDGE_WAVES * 2] = st.w[W][slot][r][0];
      }
      if (cx.seg_last) {
#pragma unroll
        for (int r = 0; r < SF_RJ; ++r)
          lds_all[SF_EDGE_IMAGE(g) + cx.edge0 + own + r * SF_EDGE_WAVES * 2 + 1] = st.w[W][slot][r][SF_VK - 1];
      }
    }
    sf_publish<W + 1, PH>(st, lds_all, cx);
  }
}

// the virtual waves beside every row hold the boundary constant the window's consumer declares
template <int W>
__device__ __forceinline__ void sf_edge_prefill(sf_t* lds_all, const sf_ctx& cx) {
  if constexpr (W < SF_NWIN) {
    if constexpr (sf_win<W>::lateral) {
      constexpr int consumer = sf_win_info<W>::consumer;
      const sf_t bc = (W < SF_T) ? sf_stage<(consumer > 0 ? consumer : 1)>::bc() : sf_stage<(consumer > 0 ? consumer : 1)>::xbc();
      if (cx.seg_first && (cx.wave == 0 || cx.wave == SF_WPR - 1)) {
#pragma unroll
        for (int image = 0; image < sf_win<W>::ring; ++image)
#pragma unroll
          for (int r = 0; r < SF_RJ; ++r) {
            sf_t* lds = lds_all + SF_EDGE_IMAGE(sf_win_base<W>::value + image);
            if (cx.wave == 0) lds[sf_edge_at(cx.ty, r, -1, 1)] = bc;
            if (cx.wave == SF_WPR - 1) lds[sf_edge_at(cx.ty, r, SF_WPR, 0)] = bc;
          }
      }
    }
    sf_edge_prefill<W + 1>(lds_all, cx);
  }
}

template <int PH>
__device__ __forceinline__ void sf_step(sf_state& st, sf_t* lds_all, sf_t* __restrict__ out, const sf_scalars& sc,
                                        const sf_ctx& cx, const int p, const int p_end) {
#if SF_OPAQUE
  // the window is made opaque at the step boundary (star3d.h): otherwise whole planes
  // of type conversions stay alive from one unrolled step to the next
#pragma unroll
  for (int w = 0; w < SF_NWIN; ++w)
#pragma unroll
    for (int s = 0; s < SF_SLOTS; ++s)
#pragma unroll
      for (int r = 0; r < SF_RJ; ++r) {
        if (s == (PH + 3) % SF_SLOTS) continue;  // in flight (loads) or not yet written
        if (s == (PH + 2) % SF_SLOTS && (w == 0 || w >= SF_T)) continue;  // a loaded plane: leave its wait where it is needed
        if constexpr (SF_VK == 1) asm volatile("" : "+v"(st.w[w][s][r][0]));
        else asm volatile("" : "+v"(st.w[w][s][r]));
      }
#endif
  sf_publish<0, PH>(st, lds_all, cx);
  __syncthreads();
  sf_nbr nbn, nbx;
  sf_gather_begin<SF_T - 1, sf_stage<SF_T>::need, PH>(nbn, st, lds_all, cx);
  if constexpr (sf_stage<SF_T>::xneed != 0)
    sf_gather_begin<(sf_stage<SF_T>::xneed != 0 ? sf_stage<SF_T>::xwin : 0), sf_stage<SF_T>::xneed, PH>(nbx, st, lds_all, cx);
  sf_stages_desc<SF_T, PH>(st, lds_all, sc, out, cx, p, p_end, nbn, nbx);
}

// first planes of an extra field's window (stage S at step p0 consumes q-1, q, q+1
// with q = p0 - (2S - 1); q+1 and q+2 are requested here, q+3 during the first step)
template <int S>
__device__ __forceinline__ void sf_extra_preload(sf_state& st, const sf_ctx& cx, const int p0) {
  if constexpr (S <= SF_T) {
    using stage = sf_stage<S>;
    if constexpr (stage::xneed != 0) {
      const int q = p0 - (2 * S - 1);
      const sf_t* field = static_cast<const sf_t*>(cx.xp.p[stage::xneed != 0 ? stage::xarg : 0]);
#pragma unroll
      for (int d = 1; d <= 2; ++d) {
        const int qx = q + d;
        const bool want = qx >= cx.cb - (SF_T - S) - 1 && qx < cx.ce + (SF_T - S) + 1;
#pragma unroll
        for (int r = 0; r < SF_RJ; ++r)
          st.w[stage::xneed != 0 ? stage::xwin : 0][d + 1][r] = sf_load_row<stage::xbc_zero>(cx, field, qx, r, want, stage::xbc());
      }
    }
    sf_extra_preload<S + 1>(st, cx, p0);
  }
}

extern "C" __global__ void __launch_bounds__(SF_BX* SF_BY)
    SF_KERNEL_NAME(const sf_t* __restrict__ in, sf_t* __restrict__ out, sf_scalars sc, sf_auxptrs xp, int halo,
                   int goff, int i_begin, int i_end, int li, int nch1, int i_begin2, int i_end2) {
  __shared__ sf_t lds_all[SF_LDS_ELEMS];

  sf_ctx cx;
  cx.in = in;
  cx.xp = xp;
  const int sf_tid_x = (int)threadIdx.x, sf_tid_y = (int)threadIdx.y;
  cx.tx = sf_tid_x;
  cx.ty = sf_tid_y;
  cx.wave = cx.tx / 64;
  cx.lane = cx.tx % 64;
  cx.seg_first = cx.lane == 0;
  cx.seg_last = cx.lane == 64 - 1;
  cx.goff = goff;
  cx.halo = halo;
  cx.row_own = sf_rows_at(cx.ty, 0) + cx.tx * SF_VK;
  cx.row_lo = sf_rows_at(cx.ty > 0 ? cx.ty - 1 : 0, 1) + cx.tx * SF_VK;
  cx.row_hi = sf_rows_at(cx.ty < SF_BY - 1 ? cx.ty + 1 : SF_BY - 1, 0) + cx.tx * SF_VK;
  cx.edge0 = (cx.ty * SF_RJ * SF_EDGE_WAVES + cx.wave) * 2;  // = sf_edge_at(ty - 1, 0, wave - 1, 0)

  // XCD-aware block order: j-adjacent tiles (sharing halo rows) land on one XCD / L2
  const int nb = gridDim.x, b = blockIdx.x;
  const int xq = nb >> 3, xr = nb & 7, xcd = b & 7;
  const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  const int jt = L % SF_NJT;
  const int kt = (L / SF_NJT) % SF_NKT;
  const int ch = L / (SF_NJT * SF_NKT);
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
  unsigned store_mask = 0;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const int j = cx.j0 + r, tr = cx.ty * SF_RJ + r;
    const bool in_dom = (j >= 0) && (j < SF_N1);
    cx.jmask |= (in_dom ? 1u : 0u) << r;
    store_mask |= ((in_dom && (SF_NOJ || (tr >= SF_T && tr < SF_TJH - SF_T))) ? 1u : 0u) << r;
  }
#pragma unroll
  for (int v = 0; v < SF_VK; ++v) cx.kmask |= ((cx.k0 + v >= 0 && cx.k0 + v < SF_N2) ? 1u : 0u) << v;
  cx.kvec_in = (cx.kmask & 1u) != 0;  // N2 % VK == 0: whole vector in or out
  if (SF_KTILED) {
    const int tk = cx.tx * SF_VK;
    if (!(tk >= SF_HK && tk < SF_TKH - SF_HK && cx.kvec_in)) store_mask = 0;
  }
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const unsigned off = (unsigned)(((cx.j0 + r) * SF_N2 + cx.k0) * (int)sizeof(sf_t));
    cx.ld_off[r] = (((cx.jmask >> r) & 1u) && cx.kvec_in) ? off : SF_OOB;
    cx.st_off[r] = ((store_mask >> r) & 1u) ? off : SF_OOB;
  }

  sf_state st;
#pragma unroll
  for (int w = 0; w < SF_NWIN; ++w)
#pragma unroll
    for (int s = 0; s < SF_SLOTS; ++s)
#pragma unroll
      for (int r = 0; r < SF_RJ; ++r) st.w[w][s][r] = (sf_vec)(sf_t)0;

  // input planes [cb - T, ce + T) are read; stage S lags 2S - 1 steps, so the last
  // stored plane ce - 1 is produced at step ce + 2T - 2
  const int p_begin = cx.cb - SF_T, p_end = cx.ce + SF_T, p_last = cx.ce + 2 * SF_T - 1;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    st.w[0][2][r] = sf_load_row<sf_stage<1>::bc_zero>(cx, in, p_begin, r, true, sf_stage<1>::bc());  // `next` of phase 0
    st.w[0][3][r] = sf_load_row<sf_stage<1>::bc_zero>(cx, in, p_begin + 1, r, p_begin + 1 < p_end, sf_stage<1>::bc());
  }
  sf_extra_preload<1>(st, cx, p_begin);
  sf_edge_prefill<0>(lds_all, cx);  // (ordered before the first reads by the first step's barrier)

  // the trip always runs four steps; up to three surplus steps past p_last compute
  // planes nobody stores (loads and stores are range-guarded)
  for (int p = p_begin; p < p_last; p += SF_SLOTS) {
    sf_step<0>(st, lds_all, out, sc, cx, p, p_end);
    sf_step<1>(st, lds_all, out, sc, cx, p + 1, p_end);
    sf_step<2>(st, lds_all, out, sc, cx, p + 2, p_end);
    sf_step<3>(st, lds_all, out, sc, cx, p + 3, p_end);
  }
}
