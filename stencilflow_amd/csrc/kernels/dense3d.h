// dense3d.h — one operator with a DENSE neighbourhood (any subset of the offsets
// {-2..2}^3 of one field: the 125-point box the reference's generator emits for an extent
// of 2, bin/synthesize.py:19-31,91-104 `box`; 25 points in 2-D) per launch, with the (j,k)
// tile of every plane staged in LDS (CDNA4 / gfx950, wave64).  Compiled at plan creation
// by hipRTC with the macros and the `sf_dense` functor emitted by codegen (gen_dense).
//
// Why not the register windows of star3d.h / compact3d.h: with 25 neighbours per plane most
// of a point's operands live in OTHER threads' registers; here every plane of the tile is in
// LDS once and every thread reads its (RJ + 2R) x (VK + 2R) patch from there -- the "2.5-D
// LDS plane tiling" of the literature.  Semantics per point as ExpandStencilCPU
// (stencilflow/stencil/cpu.py:58-115): out-of-domain reads yield the boundary constant --
// the tile is padded with it at GLOBAL coordinates when it is written to LDS; the sum is
// evaluated in the order of the program text (the functor is the text).
//
// Decomposition
//   block  = tile of TJ x TK output points of the (j,k) plane (TJ = BY * RJ rows, TK = BX * VK
//            columns), marching along i over one chunk of planes;
//   LDS    = ring of SIX plane slots of (TJ + 2R) x (TK + 2R) elements: step p writes plane p
//            into slot p mod 6 (its loads were issued a step earlier), ONE barrier, then output
//            plane q = p - R is evaluated from the five slots q-R .. q+R; the slot written at
//            step p + 1 holds plane p - 5, which nobody reads any more;
//   thread = RJ rows x VK columns of outputs; the step loop is unrolled by six, so a slot
//            index is a compile-time constant and every LDS address is one per-thread base
//            plus an immediate offset.
//
// Macros from codegen: SF_R SF_VK SF_RJ SF_BX SF_BY SF_NOJ SF_N0G SF_N1 SF_N2 SF_NJT SF_NKT SF_NT
//   SF_NLOADS SF_KERNEL_NAME; typedef sf_t; struct sf_scalars; struct sf_auxptrs;
//   struct sf_dense {bc(), bc_zero, template<int PH> apply_row(tb, r, sc, o, gi, gj, gk0)}: the VK outputs of one row,
//   every row segment (VK + 2R elements of one (di, dj)) read from LDS as aligned 16-byte chunks;
//   SF_DENSE_ROWS 1 (the operator is one plain sum): apply_rows(tb, sc, out[RJ][VK]) instead, all rows in step.
//   SF_DENSE_STREAM 1 (a plain sum whose terms come plane by plane, lowest plane first -- the generator's order, so the
//   partial sum of output plane q meets its terms in the order of the text while the planes q-R .. q+R stream past):
//   accumulate<PH>(tb, acc[5][RJ][VK]) adds the plane that has just arrived to the accumulators of the output planes
//   it belongs to, finish(sc, acc[s], out) scales the finished one.  A plane is read from LDS ONCE instead of five
//   times (1.5 instead of 7.6 ds_read_b128 per point of the 125-point box), the LDS ring shrinks to two slots, and
//   SFD_DLAST (the highest plane offset of the text) says which output plane a step completes.
//   SF_DENSE_T2 1 (round 4): TWO such operators with offsets in {-1,0,1}^3 in one launch -- `sf_dense` over the input
//   ring, its finished planes (padded with the second operator's boundary constant outside the global domain) go into a
//   second LDS ring, `sf_dense2` streams over that one a step later; three accumulator sets per operator, one barrier
//   per step.  The block evaluates both operators on its whole thread tile and stores the second one's interior (one row
//   -- and, when a row is cut into tiles, four columns -- on either side are recomputed by the neighbouring tile); no
//   register windows and no lane exchange, so the f32 adds of co-resident waves overlap.

typedef sf_t sf_vec __attribute__((ext_vector_type(SF_VK)));
typedef sf_t sf_pair __attribute__((ext_vector_type(2)));
#define SF_CE (16 / (int)sizeof(sf_t))  // elements of a 16-byte chunk
typedef sf_t sf_chunk __attribute__((ext_vector_type(16 / sizeof(sf_t))));
#ifndef SF_RC
#define SF_RC SF_R  // halo COLUMNS of an LDS row: R, rounded up to an even number (radius 3: four) -- pairs and chunks stay aligned
#endif
#define SF_SEG (SF_VK + 2 * SF_RC)  // a row segment: the thread's VK columns and RC more on either side
typedef unsigned sf_u4 __attribute__((ext_vector_type(4)));
typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));

#ifndef SF_DENSE_STREAM
#define SF_DENSE_STREAM 0
#endif
// Timing diagnostics of the streaming form (plan option debug.whatif; the results are WRONG): 1 no barrier, 2 the plane
// is not written to LDS, 8 no loads of the streamed planes, 16 results evaluated but not stored.
#ifndef SF_WHATIF
#define SF_WHATIF 0
#endif
#ifndef SF_DENSE_LOAD_EARLY
#define SF_DENSE_LOAD_EARLY 0
#endif
#ifndef SF_DENSE_T2
#define SF_DENSE_T2 0
#endif
#if SF_DENSE_STREAM
#ifndef SF_T2_ONE_IN
#define SF_T2_ONE_IN 0  // fused form: ONE slot for the input planes (a second barrier per step) instead of two
#endif
#define SF_MID0 (SF_T2_ONE_IN ? 1 : 2)  // first slot of the ring between the two operators
#define SF_SLOTS (SF_DENSE_T2 ? SF_MID0 + 2 : 2)  // LDS: the plane being read and the one being written (T2: of either ring)
#ifndef SF_ACCS
#define SF_ACCS (2 * SF_R + 1)  // accumulator sets: output planes p - R .. p + R are open while plane p is read
#endif
#else
#define SF_SLOTS 6
#endif
#define SF_OOB 0x80000000u
#define SF_PLANE_ELEMS ((long long)SF_N1 * (long long)SF_N2)
#define SF_PLANE_BYTES ((unsigned)(SF_PLANE_ELEMS * (long long)sizeof(sf_t)))
#define SF_RSRC_FLAGS 0x00020000 /* raw buffer, 32-bit data format (gfx9 / CDNA) */

#if SF_NOJ
#define SF_TJ 1
#ifndef SF_RJH
#define SF_RJH 0  // no row axis: no halo rows
#endif
#else
#define SF_TJ (SF_BY * SF_RJ)
#ifndef SF_RJH
#define SF_RJH SF_R
#endif
#endif
#ifndef SF_KTILED
#define SF_KTILED 1
#endif
#define SF_TK (SF_BX * SF_VK)
#define SF_LROWS (SF_TJ + 2 * SF_RJH)
#define SF_LS (SF_TK + 2 * SF_RC)  // row stride of a slot (elements); TK is a multiple of 4, RC even: pairs stay 8-byte aligned
#define SF_SLOT_ELEMS (SF_LROWS * SF_LS)
#define SF_PAIRS_PER_ROW (SF_LS / 2)
#define SF_PAIRS (SF_LROWS * SF_PAIRS_PER_ROW)
#define SF_THREADS (SF_BX * SF_BY)

// slot of plane q + di when the plane written this step (p = q + R) sits in slot PH
#define SF_SLOT_OF(PH, di) (((PH) + (di) - SF_R + 2 * SF_SLOTS) % SF_SLOTS)

template <typename V, int aux>
__device__ __forceinline__ V sf_buf_load(const __amdgpu_buffer_rsrc_t rs, const unsigned off) {
  if constexpr (sizeof(V) == 4) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, aux));
  } else if constexpr (sizeof(V) == 8) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, aux));
  } else {
    static_assert(sizeof(V) == 16, "4, 8 or 16 bytes");
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, aux));
  }
}
template <typename V, int aux>
__device__ __forceinline__ void sf_buf_store(const V v, const __amdgpu_buffer_rsrc_t rs, const unsigned off) {
  if constexpr (sizeof(V) == 8) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sf_u2, v), rs, off, 0, aux);
  } else if constexpr (sizeof(V) == 16) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sf_u4, v), rs, off, 0, aux);
  } else {
    static_assert(sizeof(V) == 32, "8, 16 or 32 bytes");
    struct Two { sf_u4 lo, hi; };
    const Two two = __builtin_bit_cast(Two, v);
    __builtin_amdgcn_raw_buffer_store_b128(two.lo, rs, off, 0, aux);
    __builtin_amdgcn_raw_buffer_store_b128(two.hi, rs, off + 16u, 0, aux);
  }
}

struct sf_ctx {
  const sf_t* in;
  int goff, halo, cb, ce;
  int j0, k0;  // global (j, k) of the thread's first output point (`copy` boundaries)
  // what this thread moves of every plane: SF_NLOADS pairs of elements (8 bytes for float, 16 for
  // double) -- byte offset inside the plane (SF_OOB: outside the (j,k) domain, or no pair at
  // all) and element index inside an LDS slot (-1: no pair)
  unsigned ld_off[SF_NLOADS];
  int ld_lds[SF_NLOADS];
  unsigned st_off[SF_RJ];  // byte offset of the thread's output vector in row r, or SF_OOB
  int tb;                  // LDS element index of the thread's patch origin (row -RJH, column -R of its outputs)
#if SF_DENSE_T2
  unsigned jmask, kmask;   // rows / columns of the thread's points that lie inside the global domain
#endif
};

// the thread's pairs of input plane p, padded with the boundary constant outside the global domain
__device__ __forceinline__ void sf_load_plane(const sf_ctx& cx, const int p, sf_pair (&dst)[SF_NLOADS], const bool enabled) {
  const bool plane_ok = enabled && (p + cx.goff >= 0) && (p + cx.goff < SF_N0G);
  const char* base = reinterpret_cast<const char*>(cx.in) + (long long)(p + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0,
                                                                     plane_ok ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
#pragma unroll
  for (int n = 0; n < SF_NLOADS; ++n) {
    sf_pair v = sf_buf_load<sf_pair, (SF_NT & 2) ? 2 : 0>(rs, cx.ld_off[n]);
    if constexpr (!sf_dense::bc_zero) {
      const bool ok = plane_ok && cx.ld_off[n] != SF_OOB;
      v[0] = ok ? v[0] : sf_dense::bc();
      v[1] = ok ? v[1] : sf_dense::bc();
    }
    dst[n] = v;
  }
}

#if !SF_DENSE_STREAM
// One step: plane p (in `regs`) goes to slot PH, the loads of plane p + 1 are issued, output plane
// q = p - R is evaluated and stored.
template <int PH>
__device__ __forceinline__ void sf_step(sf_t* lds, sf_pair (&regs)[SF_NLOADS], sf_t* __restrict__ out,
                                        const sf_scalars& sc, const sf_ctx& cx, const int p, const int p_end) {
#pragma unroll
  for (int n = 0; n < SF_NLOADS; ++n)
    if (cx.ld_lds[n] >= 0) *reinterpret_cast<sf_pair*>(&lds[PH * SF_SLOT_ELEMS + cx.ld_lds[n]]) = regs[n];
  __syncthreads();
  sf_load_plane(cx, p + 1, regs, p + 1 < p_end);  // lands during the evaluation below
  const int q = p - SF_R;
  const bool store_plane = q >= cx.cb && q < cx.ce && (q + cx.goff >= 0) && (q + cx.goff < SF_N0G);
  char* base = reinterpret_cast<char*>(out) + (long long)(q + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(base, 0, store_plane ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
  const sf_t* tb = lds + cx.tb;
#if SF_DENSE_ROWS
  // the operator is one plain sum: all rows of the thread are accumulated in step (a row segment read
  // from LDS serves every row that needs it), then stored
  sf_t rows[SF_RJ][SF_VK];
  sf_dense::template apply_rows<PH>(tb, sc, rows);
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) o[v] = rows[r][v];
    sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
  }
#else
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    sf_t row[SF_VK];
    sf_dense::template apply_row<PH>(tb, r, sc, row, q + cx.goff, cx.j0 + r, cx.k0);
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) o[v] = row[v];
    sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
    __builtin_amdgcn_sched_barrier(0);  // rows in order: bounds the live row segments
  }
#endif
}

#endif  // !SF_DENSE_STREAM

#if SF_DENSE_STREAM
// One step of the streaming form: plane p (in `regs`) goes to LDS slot `slot`, the loads of plane p + 1 are issued,
// the plane is added to the open output planes (set of output plane q: (q - p_begin) mod 5, so with
// PH = (p - p_begin) mod 5 the plane at offset di belongs to set (PH - di) mod 5), output plane p - SFD_DLAST is
// finished and stored.
template <int PH>
__device__ __forceinline__ void sf_step_stream(sf_t* lds, sf_pair (&regs)[SF_NLOADS], sf_t* __restrict__ out,
                                               const sf_scalars& sc, const sf_ctx& cx, const int p, const int p_end,
                                               const int slot, sf_dense::acc_t (&acc)[SF_ACCS][SF_RJ][SF_VK]) {
  sf_t* sl = lds + slot * SF_SLOT_ELEMS;
  if constexpr (!(SF_WHATIF & 2)) {
#pragma unroll
    for (int n = 0; n < SF_NLOADS; ++n)  // (only the last round of pairs can run out of pairs)
      if (n < SF_NLOADS - 1 || cx.ld_lds[n] >= 0) *reinterpret_cast<sf_pair*>(&sl[cx.ld_lds[n]]) = regs[n];
  }
#if SF_DENSE_LOAD_EARLY
  // requested before the barrier (the LDS writes above have taken their operands): the wait at the barrier is part of
  // the time the loads have to land
  if constexpr (!(SF_WHATIF & 8)) sf_load_plane(cx, p + 1, regs, p + 1 < p_end);
  if constexpr (!(SF_WHATIF & 1)) __syncthreads();  // (two slots: the waves still reading the other slot are at most one step behind)
#else
  if constexpr (!(SF_WHATIF & 1)) __syncthreads();  // (two slots: the waves still reading the other slot are at most one step behind)
  if constexpr (!(SF_WHATIF & 8)) sf_load_plane(cx, p + 1, regs, p + 1 < p_end);  // lands during the evaluation below
#endif
  sf_dense::template accumulate<PH>(sl + cx.tb, acc);
  // the sums are complete HERE: left alone, the compiler sinks the adds of an output plane towards the step that
  // finishes it and keeps the operands -- whole planes of the patch -- alive until then (244 registers instead of ~100)
#pragma unroll
  for (int a = 0; a < SF_ACCS; ++a)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) asm volatile("" : "+v"(acc[a][r][v]));
  const int q = p - SFD_DLAST;
  const bool store_plane = q >= cx.cb && q < cx.ce && (q + cx.goff >= 0) && (q + cx.goff < SF_N0G);
  char* base = reinterpret_cast<char*>(out) + (long long)(q + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(base, 0, store_plane ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
  sf_t rows[SF_RJ][SF_VK];
  sf_dense::finish(sc, acc[(PH - SFD_DLAST + 2 * SF_ACCS) % SF_ACCS], rows);
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) o[v] = rows[r][v];
    if constexpr ((SF_WHATIF & 16) != 0) asm volatile("" : : "v"(o), "s"(rs));
    else sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
  }
}
#endif

#if SF_DENSE_T2
// One step of the fused form.  PH = (p - p_begin) mod 3 names the accumulator sets as in sf_step_stream.  Operator 1
// reads input plane p from slot `s0` of ring 0 and finishes its plane q1 = p - SFD_DLAST, which goes to ring 1 (slot
// parity of q1); operator 2 reads the plane that went there in the PREVIOUS step (q1 - 1: this step's barrier has
// made it visible) and finishes output plane q1 - 1 - SFD2_DLAST.
template <int PH>
__device__ __forceinline__ void sf_step_t2(sf_t* lds, sf_pair (&regs)[SF_NLOADS], sf_t* __restrict__ out,
                                           const sf_scalars& sc, const sf_ctx& cx, const int p, const int p_load_end,
                                           const int s0, sf_dense::acc_t (&acc1)[SF_ACCS][SF_RJ][SF_VK],
                                           sf_dense2::acc_t (&acc2)[SF_ACCS][SF_RJ][SF_VK]) {
  // (requesting planes TWO steps ahead -- two register sets, the loop unrolled by six -- was measured: slower,
  //  profiles/r04_dense_t2.log)
  sf_t* in_slot = lds + (SF_T2_ONE_IN ? 0 : s0) * SF_SLOT_ELEMS;
  if constexpr (SF_T2_ONE_IN != 0) __syncthreads();  // every wave has read the plane the slot held
  if constexpr (!(SF_WHATIF & 2)) {
#pragma unroll
    for (int n = 0; n < SF_NLOADS; ++n)  // (only the last round of pairs can run out of pairs)
      if (n < SF_NLOADS - 1 || cx.ld_lds[n] >= 0) *reinterpret_cast<sf_pair*>(&in_slot[cx.ld_lds[n]]) = regs[n];
  }
  if constexpr (!(SF_WHATIF & 1)) __syncthreads();
  if constexpr (!(SF_WHATIF & 8)) sf_load_plane(cx, p + 1, regs, p + 1 < p_load_end);
  // ---- operator 1: plane p joins the open planes, plane q1 is finished and published
  sf_dense::template accumulate<PH>(in_slot + cx.tb, acc1);
#pragma unroll
  for (int a = 0; a < SF_ACCS; ++a)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) asm volatile("" : "+v"(acc1[a][r][v]));
  const int q1 = p - SFD_DLAST;
  const bool plane1_in = (q1 + cx.goff >= 0) && (q1 + cx.goff < SF_N0G);
  sf_t mid[SF_RJ][SF_VK];
  sf_dense::finish(sc, acc1[(PH - SFD_DLAST + 2 * SF_ACCS) % SF_ACCS], mid);
  const int mid_par = (s0 + 2 - SFD_DLAST) & 1;
  sf_t* mid_w = lds + (SF_MID0 + mid_par) * SF_SLOT_ELEMS + cx.tb + SF_RJH * SF_LS + SF_RC;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    // outside the global domain operator 2 reads ITS boundary constant
    const bool row_in = plane1_in && ((cx.jmask >> r) & 1u);
#pragma unroll
    for (int v = 0; v < SF_VK; v += 2) {
      sf_pair w;
      w[0] = (row_in && ((cx.kmask >> v) & 1u)) ? mid[r][v] : sf_dense2::bc();
      w[1] = (row_in && ((cx.kmask >> (v + 1)) & 1u)) ? mid[r][v + 1] : sf_dense2::bc();
      *reinterpret_cast<sf_pair*>(&mid_w[r * SF_LS + v]) = w;
    }
  }
  // ---- operator 2 on the plane published a step ago
  constexpr int PH2 = (PH - SFD_DLAST - 1 + 2 * SF_ACCS) % SF_ACCS;
  const sf_t* mid_r = lds + (SF_MID0 + (mid_par ^ 1)) * SF_SLOT_ELEMS + cx.tb;
  sf_dense2::template accumulate<PH2>(mid_r, acc2);
#pragma unroll
  for (int a = 0; a < SF_ACCS; ++a)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) asm volatile("" : "+v"(acc2[a][r][v]));
  const int q2 = q1 - 1 - SFD2_DLAST;
  const bool store_plane = q2 >= cx.cb && q2 < cx.ce && (q2 + cx.goff >= 0) && (q2 + cx.goff < SF_N0G);
  char* base = reinterpret_cast<char*>(out) + (long long)(q2 + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(base, 0, store_plane ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
  sf_t rows[SF_RJ][SF_VK];
  sf_dense2::finish(sc, acc2[(PH2 - SFD2_DLAST + 2 * SF_ACCS) % SF_ACCS], rows);
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) o[v] = rows[r][v];
    if constexpr ((SF_WHATIF & 16) != 0) asm volatile("" : : "v"(o), "s"(rs));
    else sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
  }
}
#endif

extern "C" __global__ void __launch_bounds__(SF_THREADS, 2)
    SF_KERNEL_NAME(const sf_t* __restrict__ in, sf_t* __restrict__ out, sf_scalars sc, sf_auxptrs aux, int halo,
                   int goff, int i_begin, int i_end, int li, int nch1, int i_begin2, int i_end2) {
  (void)aux;
  __shared__ sf_t lds[SF_SLOTS * SF_SLOT_ELEMS];

  sf_ctx cx;
  cx.in = in;
  cx.goff = goff;
  cx.halo = halo;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * SF_BX + tx;

  // XCD-aware block order: consecutive logical tiles (adjacent in j) land on one XCD / L2
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

#if SF_DENSE_T2
  // the thread tile overlaps its neighbours: the second operator's results are valid one row (four columns when a row
  // is cut into tiles) inside it
  const int tj0 = SF_NOJ ? 0 : jt * (SF_TJ - 2) - 1, tk0 = SF_KTILED ? kt * (SF_TK - 8) - 4 : 0;
#else
  const int tj0 = SF_NOJ ? 0 : jt * SF_TJ, tk0 = kt * SF_TK;  // first output point of the tile
#endif
  // the pairs this thread loads of every plane (row-major over the slot, pairs of columns)
#pragma unroll
  for (int n = 0; n < SF_NLOADS; ++n) {
    const int pair = tid + n * SF_THREADS;
    const int row = pair / SF_PAIRS_PER_ROW, col = (pair - row * SF_PAIRS_PER_ROW) * 2;
    const int j = tj0 - SF_RJH + row, k = tk0 - SF_RC + col;
    const bool mine = pair < SF_PAIRS;
    // (N2 is a multiple of 4 and R even or the tile origin a multiple of 4: a pair is inside or outside as a whole
    //  for R = 2; for R = 1 the pair straddles the edge -- R is always 2 here, codegen enforces it)
    const bool inside = mine && j >= 0 && j < SF_N1 && k >= 0 && k + 1 < SF_N2;
    cx.ld_off[n] = inside ? (unsigned)((j * SF_N2 + k) * (int)sizeof(sf_t)) : SF_OOB;
    cx.ld_lds[n] = mine ? row * SF_LS + col : -1;
  }
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const int j = tj0 + (SF_NOJ ? 0 : ty * SF_RJ + r), k = tk0 + tx * SF_VK;
#if SF_DENSE_T2
    const int tr = ty * SF_RJ + r, tc = tx * SF_VK;
    const bool valid = (SF_NOJ || (tr >= 1 && tr < SF_TJ - 1)) && (!SF_KTILED || (tc >= 4 && tc < SF_TK - 4));
    const bool inside = valid && j >= 0 && j < SF_N1 && k >= 0 && k + SF_VK <= SF_N2;
#else
    const bool inside = j < SF_N1 && k + SF_VK <= SF_N2;
#endif
    cx.st_off[r] = inside ? (unsigned)((j * SF_N2 + k) * (int)sizeof(sf_t)) : SF_OOB;
  }
  cx.tb = (SF_NOJ ? 0 : ty * SF_RJ) * SF_LS + tx * SF_VK;
  cx.j0 = tj0 + (SF_NOJ ? 0 : ty * SF_RJ);
  cx.k0 = tk0 + tx * SF_VK;
#if SF_DENSE_T2
  cx.jmask = 0;
  cx.kmask = 0;
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) cx.jmask |= ((SF_NOJ || (cx.j0 + r >= 0 && cx.j0 + r < SF_N1)) ? 1u : 0u) << r;
#pragma unroll
  for (int v = 0; v < SF_VK; ++v) cx.kmask |= ((cx.k0 + v >= 0 && cx.k0 + v < SF_N2) ? 1u : 0u) << v;
#endif

  // input planes [p_begin, p_end) are read; step p writes plane p into slot (p - p_begin) mod 6
  const int p_begin = cx.cb - SF_R, p_end = cx.ce + SF_R;
  sf_pair regs[SF_NLOADS];
  sf_load_plane(cx, p_begin, regs, true);
  // the slots of planes before p_begin are never read for a stored plane: the first stored plane is
  // cb = p_begin + R, whose oldest operand plane is p_begin
#if SF_DENSE_T2
  // ring 1 starts out as the second operator's boundary constant: its halo rows and columns are never written again
  // (they are right where the tile touches the edge of the domain; elsewhere the results that read them are not stored)
  for (int i = tid; i < 2 * SF_SLOT_ELEMS; i += SF_THREADS) lds[SF_MID0 * SF_SLOT_ELEMS + i] = sf_dense2::bc();
  sf_dense::acc_t acc1[SF_ACCS][SF_RJ][SF_VK];
  sf_dense2::acc_t acc2[SF_ACCS][SF_RJ][SF_VK];
#pragma unroll
  for (int a = 0; a < SF_ACCS; ++a)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) {
        acc1[a][r][v] = (sf_dense::acc_t)0;
        acc2[a][r][v] = (sf_dense2::acc_t)0;
      }
  // output plane q2 leaves at step q2 + SFD_DLAST + 1 + SFD2_DLAST; input planes up to ce + R - 1 are read
  const int p_stop = cx.ce + SFD_DLAST + 1 + SFD2_DLAST;
  for (int p = p_begin; p < p_stop; p += SF_ACCS) {
    const int s0 = (p - p_begin) & 1;  // three steps per trip: the slot parity alternates from trip to trip
    sf_step_t2<0>(lds, regs, out, sc, cx, p, p_end, s0, acc1, acc2);
    sf_step_t2<1>(lds, regs, out, sc, cx, p + 1, p_end, s0 ^ 1, acc1, acc2);
    sf_step_t2<2>(lds, regs, out, sc, cx, p + 2, p_end, s0, acc1, acc2);
  }
#elif SF_DENSE_STREAM
  // (an output plane before cb collects planes that were never added to it: it is not stored; the first stored plane
  //  cb = p_begin + R opens at step p_begin + R + d0 >= p_begin, with the first term of the text -- an assignment)
  sf_dense::acc_t acc[SF_ACCS][SF_RJ][SF_VK];
#pragma unroll
  for (int a = 0; a < SF_ACCS; ++a)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) acc[a][r][v] = (sf_dense::acc_t)0;
  for (int p = p_begin; p < p_end; p += SF_ACCS) {
    const int s0 = (p - p_begin) & 1;  // an odd number of steps per trip: the slot parity alternates from trip to trip
    sf_step_stream<0>(lds, regs, out, sc, cx, p, p_end, s0, acc);
    sf_step_stream<1>(lds, regs, out, sc, cx, p + 1, p_end, s0 ^ 1, acc);
    sf_step_stream<2>(lds, regs, out, sc, cx, p + 2, p_end, s0, acc);
    sf_step_stream<3>(lds, regs, out, sc, cx, p + 3, p_end, s0 ^ 1, acc);
    sf_step_stream<4>(lds, regs, out, sc, cx, p + 4, p_end, s0, acc);
#if SF_ACCS == 7
    sf_step_stream<5>(lds, regs, out, sc, cx, p + 5, p_end, s0 ^ 1, acc);
    sf_step_stream<6>(lds, regs, out, sc, cx, p + 6, p_end, s0, acc);
#endif
  }
#else
  for (int p = p_begin; p < p_end; p += SF_SLOTS) {
    sf_step<0>(lds, regs, out, sc, cx, p, p_end);
    sf_step<1>(lds, regs, out, sc, cx, p + 1, p_end);
    sf_step<2>(lds, regs, out, sc, cx, p + 2, p_end);
    sf_step<3>(lds, regs, out, sc, cx, p + 3, p_end);
    sf_step<4>(lds, regs, out, sc, cx, p + 4, p_end);
    sf_step<5>(lds, regs, out, sc, cx, p + 5, p_end);
  }
#endif
}
