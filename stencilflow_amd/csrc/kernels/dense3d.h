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
// the tile is padded with it at GLOBAL coordinates in LDS; the sum is evaluated in the order
// of the program text (the functor is the text).
//
// Decomposition
//   block  = tile of TJ x TK output points of the (j,k) plane (TJ = BY * RJ rows, TK = BX * VK
//            columns), marching along i over one chunk of planes;
//   thread = RJ rows x VK columns of outputs; the step loop is unrolled, so every LDS address is one
//            per-thread base plus an immediate offset.
// Three forms (macros from codegen):
//   * general (any expression over the neighbourhood; SF_DENSE_ROWS 1: one plain sum, rows in step): a ring of SIX
//     plane slots of (TJ + 2R) x (TK + 2R) elements: step p writes plane p -- loaded a step earlier into registers --
//     into slot p mod 6, ONE barrier, then output plane q = p - R is evaluated from the five slots q-R .. q+R.
//   * SF_DENSE_STREAM 1 (a plain sum whose terms come plane by plane, lowest plane first -- the generator's order, so the
//     partial sum of output plane q meets its terms in the order of the text while the planes q-R .. q+R stream past):
//     accumulate<PH>(tb, acc[ACCS][RJ][VK]) adds the plane that has just arrived to the accumulators of the output
//     planes it belongs to, finish(sc, acc[s], out) scales the finished one.  A plane is read from LDS ONCE instead of
//     five times; SFD_DLAST (the highest plane offset of the text) says which output plane a step completes.
//   * SF_DENSE_T2 1 (round 4): TWO such operators with offsets in {-1,0,1}^3 in one launch -- `sf_dense` over the input
//     ring, its finished planes (padded with the second operator's boundary constant outside the global domain) go into
//     a second LDS ring, `sf_dense2` streams over that one a step later; three accumulator sets per operator.  The block
//     evaluates both operators on its whole thread tile and stores the second one's interior (one row -- and, when a row
//     is cut into tiles, four columns -- on either side are recomputed by the neighbouring tile); no register windows
//     and no lane exchange, so the f32 adds of co-resident waves overlap.
//     Round 5, SF_RS 2: two operators that each reach TWO points (sums of few terms -- the generator's radius-2 crosses,
//     whose in-plane terms join their output plane two steps after their own plane arrived): five accumulator sets per
//     operator, both rings keep the planes the late terms read (SF_LAG, SF_LAG2), tiles overlap by two rows, and the ring
//     between the operators holds the TJ rows of the thread tile only (SF_MID_HALO 0).  Rows of 34 threads (136 columns,
//     128 kept: a row of 512 is four tiles) make blocks that are not whole waves: the last wave has lanes off and
//     requests no pieces (SF_ODD_WAVE).
// Round 5: the streaming forms take their planes by LDS-DMA (`buffer_load_dwordx4 ... lds`: memory -> LDS, no staging
// registers, no ds_write).  The input planes go through a ring of SF_IN_SLOTS slots: SF_LAG planes kept behind the
// one that has just arrived (terms that join their output plane late), the rest requested ahead.  One wave-instruction writes 64 x 16 bytes of LDS in a row
// (M0 + 16 x lane) while the SOURCE address is per lane: a slot is filled in image order, chunk c of the slot by lane
// c mod 64 of piece c / 64, and a lane whose chunk lies outside the (j,k) domain -- or every lane, for a plane outside
// the global domain: a resource of zero records -- reads out of range, which WRITES ZERO and touches no memory
// (measured, tools/micro/lds_dma_probe.hip, profiles/r05_lds_dma_probe.log).  So an LDS row starts SF_RCL columns left
// of the tile, RCL a whole number of chunks (a chunk is in or out as a whole: N2 is a multiple of 4), a slot is padded
// to a whole number of 1-KiB pieces, and a boundary constant other than zero is written over the zeros by the tiles
// that reach beyond the domain (sf_fix_boundary).  A row segment then starts RCL - RC elements past a 16-byte boundary
// and is read in aligned pieces (float, RC = 2: 8 + 16 + 8 bytes).
//
// Macros from codegen: SF_R SF_VK SF_RJ SF_BX SF_BY SF_NOJ SF_N0G SF_N1 SF_N2 SF_NJT SF_NKT SF_NT SF_KERNEL_NAME,
//   general forms SF_NLOADS, streaming forms SF_IN_SLOTS SF_RCL [SF_RC SF_ACCS SF_RJH SF_KTILED SF_LAG; fused: SF_RS
//   SF_LAG2 SF_MID_SLOTS SF_MID_HALO]; typedef sf_t; struct
//   sf_scalars; struct sf_auxptrs; struct sf_dense (and sf_dense2) {bc(), bc_zero, apply_row / apply_rows / accumulate + finish}.

typedef sf_t sf_vec __attribute__((ext_vector_type(SF_VK)));
typedef sf_t sf_pair __attribute__((ext_vector_type(2)));
#define SF_CE (16 / (int)sizeof(sf_t))  // elements of a 16-byte chunk
typedef sf_t sf_chunk __attribute__((ext_vector_type(16 / sizeof(sf_t))));
typedef unsigned sf_u4 __attribute__((ext_vector_type(4)));
typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));
#ifndef SF_RC
#define SF_RC SF_R  // halo COLUMNS of a row segment: R, rounded up to an even number (radius 3: four)
#endif
#ifndef SF_IN_SLOTS
#define SF_IN_SLOTS 0
#endif
#define SF_DMA (SF_IN_SLOTS > 0)
#ifndef SF_RCL
#define SF_RCL SF_RC  // halo columns of an LDS ROW (>= RC; LDS-DMA: a whole number of chunks)
#endif
#ifndef SF_DENSE_STREAM
#define SF_DENSE_STREAM 0
#endif
#ifndef SF_DENSE_T2
#define SF_DENSE_T2 0
#endif
#if SF_DENSE_STREAM
#ifndef SF_MID_SLOTS
#define SF_MID_SLOTS 2
#endif
#ifndef SF_RS
#define SF_RS 1  // fused form: what ONE of the two operators reaches (planes and rows; SF_R = 2 RS is the pair's)
#endif
#ifndef SF_LAG2
#define SF_LAG2 0  // fused form: planes the ring between the operators keeps for the second operator's late terms
#endif
#ifndef SF_NST
#define SF_NST 2  // fused form: operators per launch (three: a second ring, between the second and the third)
#endif
#ifndef SF_LAG3
#define SF_LAG3 0
#endif
#ifndef SF_MID2_SLOTS
#define SF_MID2_SLOTS (SF_NST == 3 ? 2 + SF_LAG3 : 0)
#endif
#define SF_SLOTS (SF_DENSE_T2 ? SF_IN_SLOTS + SF_MID_SLOTS : SF_IN_SLOTS)
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
#define SF_THREADS (SF_BX * SF_BY)
#define SF_LROWS (SF_TJ + 2 * SF_RJH)
#define SF_LS (SF_TK + 2 * SF_RCL)  // row stride of a slot (elements); TK is a multiple of 4, RCL even: pairs stay 8-byte aligned
#define SF_SLOT_ELEMS (SF_LROWS * SF_LS)
#if SF_DMA
#define SF_NCH (SF_SLOT_ELEMS / SF_CE)       // 16-byte chunks of a slot
#define SF_NI ((SF_NCH + 63) / 64)           // pieces (wave-instructions, 1 KiB each) that fill one
#define SF_SLOT_STRIDE (SF_NI * 64 * SF_CE)  // elements from slot to slot: the last piece ends inside the slot's padding
#define SF_NW (SF_THREADS / 64)               // whole waves of the block: the ones that request pieces (a block whose
#define SF_ODD_WAVE (SF_THREADS % 64 != 0)   // thread count is no multiple of 64 has one more wave, with lanes off)
#define SF_ND ((SF_NI + SF_NW - 1) / SF_NW)  // pieces a wave issues per plane, at most / at least
#define SF_NDMIN (SF_NI / SF_NW)
#ifndef SF_LAG
#define SF_LAG 0  // planes the ring keeps behind the one that has just arrived (terms that join their output plane late)
#endif
#define SF_AHEAD (SF_IN_SLOTS - SF_LAG - 1)  // planes requested ahead of the one being read (>= 1)
#else
#define SF_SLOT_STRIDE SF_SLOT_ELEMS
#define SF_PAIRS_PER_ROW (SF_LS / 2)
#define SF_PAIRS (SF_LROWS * SF_PAIRS_PER_ROW)
#endif
#if SF_DENSE_T2
// The ring between the operators.  SF_MID_HALO 1: slots of the input ring's geometry (halo rows that stay the second
// operator's boundary constant).  0 (the pairs of reach two): the TJ rows of the thread tile only -- a result that is
// stored reads nothing beyond them, the tiles overlapping by the second operator's reach; the threads at the rim of
// the tile, whose results are not stored, read the neighbouring slots (SF_MID_PAD keeps the last one's reads inside).
#ifndef SF_MID_HALO
#define SF_MID_HALO 1
#endif
#define SF_MID_STRIDE (SF_MID_HALO ? SF_SLOT_STRIDE : (SF_TJ * SF_LS + SF_CE - 1) / SF_CE * SF_CE)
#define SF_MID_PAD (SF_MID_HALO ? 0 : SF_RJH * SF_LS + 2 * SF_CE)
// (SF_RCL 0 -- rows without halo columns, below -- : the first thread of a slot's first row reads SF_RC elements before
//  it; one 1-KiB piece in front of slot 0 keeps that inside the array and the slots on whole pieces)
#define SF_LDS_FRONT (SF_RCL < SF_RC ? 1024 / (int)sizeof(sf_t) : 0)
#define SF_EDGE ((SF_NST - 1) * SF_RS)  // rows on either side of the thread tile whose results are not stored
#define SF_MID0 (SF_IN_SLOTS * SF_SLOT_STRIDE)              // first element of the ring between operators 1 and 2 (from slot 0)
#define SF_MIDB0 (SF_MID0 + SF_MID_SLOTS * SF_MID_STRIDE)  // ... between operators 2 and 3
#define SF_LDS_ELEMS (SF_LDS_FRONT + SF_MIDB0 + SF_MID2_SLOTS * SF_MID_STRIDE + SF_MID_PAD)
#else
#define SF_LDS_FRONT 0
#define SF_LDS_ELEMS (SF_SLOTS * SF_SLOT_STRIDE)
#endif

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
#if SF_DMA
  // what this thread requests of every plane: chunk (n * NW + wave) * 64 + lane of the slot, n < ND -- its byte offset
  // inside the plane (SF_OOB: outside the (j,k) domain or beyond the slot: reads as zero)
  unsigned ld_off[SF_ND];
  unsigned wave;      // (scalar) wave of the block
  unsigned lds_base;  // (scalar) LDS byte address of slot 0
  bool edge_tile;     // (scalar) the tile reaches beyond the (j,k) domain: non-zero boundary constants are written in
#else
  // what this thread moves of every plane: SF_NLOADS pairs of elements (8 bytes for float, 16 for
  // double) -- byte offset inside the plane (SF_OOB: outside the (j,k) domain, or no pair at
  // all) and element index inside an LDS slot (-1: no pair)
  unsigned ld_off[SF_NLOADS];
  int ld_lds[SF_NLOADS];
#endif
  unsigned st_off[SF_RJ];  // byte offset of the thread's output vector in row r, or SF_OOB
  int tb;                  // LDS element index of the thread's patch origin (row -RJH, column -RC of its outputs)
#if SF_DENSE_T2
  unsigned jmask, kmask;   // rows / columns of the thread's points that lie inside the global domain
#endif
};

#if SF_DMA
// One piece: 64 lanes x 16 bytes from `voff` (per lane, bytes inside the resource) to LDS [lds_byte + 16 * lane].  M0 holds
// the destination and belongs to the compiler: saved and restored inside the statement.  The compiler does not count
// this load: the waits below are ours (sf_wait_plane).
__device__ __forceinline__ void sf_dma16(const __amdgpu_buffer_rsrc_t rs, const unsigned voff, const unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rs), "s"(lds_byte)
      : "memory");
}

// Requests input plane p into slot `slot`.  A plane outside the global domain, or one the chunk does not need
// (`enabled` false), is requested from a resource of zero records: every lane reads zero, nothing touches memory, and
// EVERY wave issues the same number of pieces in every step -- which is what the counted waits rely on.
__device__ __forceinline__ void sf_dma_plane(const sf_ctx& cx, const int p, const bool enabled, const int slot) {
  const bool plane_ok = enabled && (p + cx.goff >= 0) && (p + cx.goff < SF_N0G);
  const char* base = reinterpret_cast<const char*>(cx.in) + (long long)(p + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0,
                                                                     plane_ok ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
  const unsigned dst = cx.lds_base + (unsigned)slot * (unsigned)(SF_SLOT_STRIDE * sizeof(sf_t));
#pragma unroll
  for (int n = 0; n < SF_ND; ++n) {
    const unsigned piece = (unsigned)n * SF_NW + cx.wave;
    if ((n < SF_NDMIN || piece < (unsigned)SF_NI) && (!SF_ODD_WAVE || cx.wave < (unsigned)SF_NW))
      sf_dma16(rs, cx.ld_off[n], dst + piece * 1024u);
  }
}

// Waits until the wave's requests of the plane that is read next have landed (all but the YOUNGER vector-memory
// operations: the pieces of the planes requested since -- at least NDMIN each -- and the stores of the steps since),
// and for its own LDS accesses; then the block's barrier.  `drain`: the first steps of a chunk, where fewer
// operations are in flight than the count assumes.
template <int YOUNGER>
__device__ __forceinline__ void sf_wait_plane(const bool drain) {
  constexpr int N = YOUNGER > 63 ? 63 : YOUNGER;
  if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// A boundary constant other than zero: the tiles that reach beyond the (j,k) domain, and every tile for a plane outside
// the global domain, write it over the zeros the requests left (each thread over the chunks it requested), then a
// second barrier.  The condition is the same for the whole block.
template <typename F>
__device__ __forceinline__ void sf_fix_boundary(const sf_ctx& cx, const int p, const bool enabled, sf_t* sl) {
  if constexpr (!F::bc_zero) {
    const bool plane_in = enabled && (p + cx.goff >= 0) && (p + cx.goff < SF_N0G);
    if (cx.edge_tile || !plane_in) {
      sf_chunk fill;
#pragma unroll
      for (int e = 0; e < SF_CE; ++e) fill[e] = F::bc();
#pragma unroll
      for (int n = 0; n < SF_ND; ++n) {
        const unsigned c = ((unsigned)n * SF_NW + cx.wave) * 64u + (threadIdx.x + threadIdx.y * SF_BX) % 64u;
        if (c < (unsigned)SF_NCH && (!plane_in || cx.ld_off[n] == SF_OOB) && (!SF_ODD_WAVE || cx.wave < (unsigned)SF_NW))
          *reinterpret_cast<sf_chunk*>(&sl[c * SF_CE]) = fill;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
}
#else
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
#endif

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
#if !SF_DMA
#error "the streaming forms take their planes by LDS-DMA (SF_IN_SLOTS >= 1)"
#endif
// the accumulators are complete HERE: left alone, the compiler sinks the adds of an output plane towards the step that
// finishes it and keeps the operands -- whole planes of the patch -- alive until then (244 registers instead of ~100)
template <typename A>
__device__ __forceinline__ void sf_pin(A (&acc)[SF_ACCS][SF_RJ][SF_VK]) {
#pragma unroll
  for (int a = 0; a < SF_ACCS; ++a)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) asm volatile("" : "+v"(acc[a][r][v]));
}

__device__ __forceinline__ void sf_store_rows(const sf_t (&rows)[SF_RJ][SF_VK], sf_t* __restrict__ out, const sf_ctx& cx, const int q) {
  const bool store_plane = q >= cx.cb && q < cx.ce && (q + cx.goff >= 0) && (q + cx.goff < SF_N0G);
  char* base = reinterpret_cast<char*>(out) + (long long)(q + cx.halo) * (long long)SF_PLANE_BYTES;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(base, 0, store_plane ? SF_PLANE_BYTES : 0u, SF_RSRC_FLAGS);
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {  // (always SF_RJ store instructions per step: the counted waits rely on it)
    sf_vec o;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) o[v] = rows[r][v];
    sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);
  }
}

#endif  // SF_DENSE_STREAM

#if SF_DENSE_STREAM && !SF_DENSE_T2
// One step of the streaming form.  Input plane p has been requested SF_AHEAD steps ago into slot `slot`; when it has
// landed everywhere (wait, barrier) the slot that held plane p - 1 is free and plane p + SF_AHEAD is requested into it;
// plane p is added to the open output planes (set of output plane q: (q - p_begin) mod ACCS, so with
// PH = (p - p_begin) mod ACCS the plane at offset di belongs to set (PH - di) mod ACCS), output plane p - SFD_DLAST is
// finished and stored.
template <int PH>
__device__ __forceinline__ void sf_step_stream(sf_t* lds, sf_t* __restrict__ out, const sf_scalars& sc, const sf_ctx& cx,
                                               const int p, const int p_begin, const int p_end, const int slot,
                                               sf_dense::acc_t (&acc)[SF_ACCS][SF_RJ][SF_VK]) {
  // younger than plane p's pieces: the stores of the AHEAD steps since, the pieces of the AHEAD - 1 planes since
  sf_wait_plane<(SF_AHEAD - 1) * SF_NDMIN + SF_AHEAD * SF_RJ>(p - p_begin < SF_AHEAD);
  sf_t* sl = lds + slot * SF_SLOT_STRIDE;
  sf_fix_boundary<sf_dense>(cx, p, p < p_end, sl);
  sf_dma_plane(cx, p + SF_AHEAD, p + SF_AHEAD < p_end, (slot + SF_AHEAD) % SF_IN_SLOTS);
  const sf_t* tb[SF_LAG + 1];  // the thread's patch of planes p, p - 1, .. p - LAG
#pragma unroll
  for (int l = 0; l <= SF_LAG; ++l) tb[l] = lds + ((slot + SF_IN_SLOTS - l) % SF_IN_SLOTS) * SF_SLOT_STRIDE + cx.tb;
  sf_dense::template accumulate<PH>(tb, sc, acc);
  sf_pin(acc);
  sf_t rows[SF_RJ][SF_VK];
  sf_dense::finish(sc, acc[(PH - SFD_DLAST + 2 * SF_ACCS) % SF_ACCS], rows);
  sf_store_rows(rows, out, cx, p - SFD_DLAST);
}
#endif

#if SF_DENSE_T2
// One step of the fused form.  PH = (p - p_begin) mod ACCS names the accumulator sets as in sf_step_stream.  Operator 1
// reads input plane p (and the SF_LAG planes before it) from the input ring and finishes its plane q1 = p - SFD_DLAST,
// which goes to slot `mw` of the ring between the operators; operator 2 reads the plane that went there in the PREVIOUS
// step (q1 - 1: this step's barrier has made it visible; and the SF_LAG2 planes before it) and finishes plane
// q2 = q1 - 1 - SFD2_DLAST -- the launch's output, or (SF_NST 3) the plane that goes to slot `mw2` of a second such ring,
// from which operator 3 finishes output plane q2 - 1 - SFD3_DLAST the same way.  The input planes are requested one
// step ahead; a ring between two operators has 2 + LAG slots, or ONE (SF_MID_SLOTS 1, two operators) where 160 KB of
// LDS hold three slots in all -- 18-row tiles of 512 columns.
template <typename F>
__device__ __forceinline__ void sf_publish(const sf_ctx& cx, const sf_t (&rows)[SF_RJ][SF_VK], const bool plane_in, sf_t* dst) {
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    // outside the global domain the next operator reads ITS boundary constant
    const bool row_in = plane_in && ((cx.jmask >> r) & 1u);
    sf_vec w;
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) w[v] = (row_in && ((cx.kmask >> v) & 1u)) ? rows[r][v] : F::bc();
    *reinterpret_cast<sf_vec*>(&dst[r * SF_LS]) = w;  // (the thread's own columns: 16-byte aligned)
  }
}

template <int PH>
__device__ __forceinline__ void sf_step_t2(sf_t* lds, sf_t* __restrict__ out, const sf_scalars& sc, const sf_ctx& cx,
                                           const int p, const int p_begin, const int p_end, const int slot, const int mw,
                                           const int mw2, sf_dense::acc_t (&acc1)[SF_ACCS][SF_RJ][SF_VK],
                                           sf_dense2::acc_t (&acc2)[SF_ACCS][SF_RJ][SF_VK], sf_dense::acc_t (&carry1)[SF_RJ][SF_VK],
                                           sf_dense2::acc_t (&carry2)[SF_RJ][SF_VK]
#if SF_NST == 3
                                           , sf_dense3::acc_t (&acc3)[SF_ACCS][SF_RJ][SF_VK], sf_dense3::acc_t (&carry3)[SF_RJ][SF_VK]
#endif
) {
  sf_wait_plane<(SF_AHEAD - 1) * SF_NDMIN + SF_AHEAD * SF_RJ>(p - p_begin < SF_AHEAD);
  sf_t* in_slot = lds + slot * SF_SLOT_STRIDE;
  sf_fix_boundary<sf_dense>(cx, p, p < p_end, in_slot);
  sf_dma_plane(cx, p + SF_AHEAD, p + SF_AHEAD < p_end, (slot + SF_AHEAD) % SF_IN_SLOTS);
  const int q1 = p - SFD_DLAST;
  constexpr int PH2 = (PH - SFD_DLAST - 1 + 2 * SF_ACCS) % SF_ACCS;
  // the thread's patch in a slot between the operators (row 0 of such a slot: the tile's row -RJH, or its row 0)
  const int mid_tb = cx.tb - (SF_MID_HALO ? 0 : SF_RJH * SF_LS);
  sf_t* mid_w = lds + SF_MID0 + mw * SF_MID_STRIDE + mid_tb + SF_RJH * SF_LS + SF_RC;
  auto second = [&]() {
    // ---- operator 2 on the plane published a step ago
    const sf_t* tb2[SF_LAG2 + 1];
#pragma unroll
    for (int l = 0; l <= SF_LAG2; ++l)
      tb2[l] = lds + SF_MID0 + (SF_MID_SLOTS == 1 ? 0 : (mw + SF_MID_SLOTS - 1 - l) % SF_MID_SLOTS) * SF_MID_STRIDE + mid_tb;
    sf_dense2::template accumulate<PH2>(tb2, sc, acc2, carry2);
    sf_pin(acc2);
    sf_t rows[SF_RJ][SF_VK];
    sf_dense2::finish(sc, acc2[(PH2 - SFD2_DLAST + 2 * SF_ACCS) % SF_ACCS], rows);
    const int q2 = q1 - 1 - SFD2_DLAST;
#if SF_NST == 3
    // ---- ... published in turn; operator 3 on the plane that went there a step ago
    sf_publish<sf_dense3>(cx, rows, (q2 + cx.goff >= 0) && (q2 + cx.goff < SF_N0G),
                          lds + SF_MIDB0 + mw2 * SF_MID_STRIDE + mid_tb + SF_RJH * SF_LS + SF_RC);
    constexpr int PH3 = (PH2 - SFD2_DLAST - 1 + 2 * SF_ACCS) % SF_ACCS;
    const sf_t* tb3[SF_LAG3 + 1];
#pragma unroll
    for (int l = 0; l <= SF_LAG3; ++l) tb3[l] = lds + SF_MIDB0 + ((mw2 + SF_MID2_SLOTS - 1 - l) % SF_MID2_SLOTS) * SF_MID_STRIDE + mid_tb;
    sf_dense3::template accumulate<PH3>(tb3, sc, acc3, carry3);
    sf_pin(acc3);
    sf_t rows3[SF_RJ][SF_VK];
    sf_dense3::finish(sc, acc3[(PH3 - SFD3_DLAST + 2 * SF_ACCS) % SF_ACCS], rows3);
    sf_store_rows(rows3, out, cx, q2 - 1 - SFD3_DLAST);
#else
    sf_store_rows(rows, out, cx, q2);
#endif
  };
  // (one slot between the operators: operator 2 reads it FIRST -- what the previous step published --, operator 1's
  //  plane goes there at the very end of the step, behind a second barrier that sits right before the next step's)
  if constexpr (SF_MID_SLOTS == 1) second();
  // ---- operator 1: plane p joins the open planes, plane q1 is finished and published
  const sf_t* tb1[SF_LAG + 1];
#pragma unroll
  for (int l = 0; l <= SF_LAG; ++l) tb1[l] = lds + ((slot + SF_IN_SLOTS - l) % SF_IN_SLOTS) * SF_SLOT_STRIDE + cx.tb;
  sf_dense::template accumulate<PH>(tb1, sc, acc1, carry1);
  sf_pin(acc1);
  sf_t mid[SF_RJ][SF_VK];
  sf_dense::finish(sc, acc1[(PH - SFD_DLAST + 2 * SF_ACCS) % SF_ACCS], mid);
  if constexpr (SF_MID_SLOTS == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read the slot
  sf_publish<sf_dense2>(cx, mid, (q1 + cx.goff >= 0) && (q1 + cx.goff < SF_N0G), mid_w);
  if constexpr (SF_MID_SLOTS != 1) second();
}
#endif

extern "C" __global__ void __launch_bounds__(SF_THREADS, 2)
    SF_KERNEL_NAME(const sf_t* __restrict__ in, sf_t* __restrict__ out, sf_scalars sc, sf_auxptrs aux, int halo,
                   int goff, int i_begin, int i_end, int li, int nch1, int i_begin2, int i_end2) {
  (void)aux;
  __shared__ __attribute__((aligned(1024))) sf_t lds_all[SF_LDS_ELEMS];
  sf_t* const lds = lds_all + SF_LDS_FRONT;  // slot 0

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
  // the thread tile overlaps its neighbours: the last operator's results are valid (NST - 1) RS rows (four columns when
  // a row is cut into tiles) inside it
  const int tj0 = SF_NOJ ? 0 : jt * (SF_TJ - 2 * SF_EDGE) - SF_EDGE, tk0 = SF_KTILED ? kt * (SF_TK - 8) - 4 : 0;
#else
  const int tj0 = SF_NOJ ? 0 : jt * SF_TJ, tk0 = kt * SF_TK;  // first output point of the tile
#endif
#if SF_DMA
  // the chunks this thread requests of every plane (image order over the slot)
  cx.wave = __builtin_amdgcn_readfirstlane((unsigned)tid >> 6);
  cx.lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
  cx.edge_tile = (tj0 - SF_RJH < 0) || (tj0 - SF_RJH + SF_LROWS > SF_N1) || (tk0 - SF_RCL < 0) || (tk0 - SF_RCL + SF_LS > SF_N2);
#pragma unroll
  for (int n = 0; n < SF_ND; ++n) {
    const int c = (n * SF_NW + (tid >> 6)) * 64 + (tid & 63);
    const int row = c / (SF_LS / SF_CE), col = (c - row * (SF_LS / SF_CE)) * SF_CE;
    const int j = tj0 - SF_RJH + row, k = tk0 - SF_RCL + col;
    // (N2 and the tile origin are multiples of the chunk: a chunk is inside or outside as a whole)
    const bool inside = c < SF_NCH && j >= 0 && j < SF_N1 && k >= 0 && k + SF_CE <= SF_N2;
    cx.ld_off[n] = inside ? (unsigned)((j * SF_N2 + k) * (int)sizeof(sf_t)) : SF_OOB;
  }
#else
  // the pairs this thread loads of every plane (row-major over the slot, pairs of columns)
#pragma unroll
  for (int n = 0; n < SF_NLOADS; ++n) {
    const int pair = tid + n * SF_THREADS;
    const int row = pair / SF_PAIRS_PER_ROW, col = (pair - row * SF_PAIRS_PER_ROW) * 2;
    const int j = tj0 - SF_RJH + row, k = tk0 - SF_RC + col;
    const bool mine = pair < SF_PAIRS;
    // (N2 is a multiple of 4 and R even or the tile origin a multiple of 4: a pair is inside or outside as a whole)
    const bool inside = mine && j >= 0 && j < SF_N1 && k >= 0 && k + 1 < SF_N2;
    cx.ld_off[n] = inside ? (unsigned)((j * SF_N2 + k) * (int)sizeof(sf_t)) : SF_OOB;
    cx.ld_lds[n] = mine ? row * SF_LS + col : -1;
  }
#endif
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r) {
    const int j = tj0 + (SF_NOJ ? 0 : ty * SF_RJ + r), k = tk0 + tx * SF_VK;
#if SF_DENSE_T2
    const int tr = ty * SF_RJ + r, tc = tx * SF_VK;
    const bool valid = (SF_NOJ || (tr >= SF_EDGE && tr < SF_TJ - SF_EDGE)) && (!SF_KTILED || (tc >= 4 && tc < SF_TK - 4));
    const bool inside = valid && j >= 0 && j < SF_N1 && k >= 0 && k + SF_VK <= SF_N2;
#else
    const bool inside = j < SF_N1 && k + SF_VK <= SF_N2;
#endif
    cx.st_off[r] = inside ? (unsigned)((j * SF_N2 + k) * (int)sizeof(sf_t)) : SF_OOB;
  }
  cx.tb = (SF_NOJ ? 0 : ty * SF_RJ) * SF_LS + tx * SF_VK + (SF_RCL - SF_RC);
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

  // input planes [p_begin, p_end) are read
  const int p_begin = cx.cb - SF_R, p_end = cx.ce + SF_R;
  // the slots of planes before p_begin are never read for a stored plane: the first stored plane is
  // cb = p_begin + R, whose oldest operand plane is p_begin
#if SF_DENSE_T2
  // the ring between the operators starts out as the second operator's boundary constant: its halo rows and columns
  // are never written again (they are right where the tile touches the edge of the domain; elsewhere the results
  // that read them are not stored)
#if SF_NST == 3
  for (int i = tid; i < SF_MID_SLOTS * SF_MID_STRIDE; i += SF_THREADS) lds[SF_MID0 + i] = sf_dense2::bc();
  for (int i = tid; i < SF_MID2_SLOTS * SF_MID_STRIDE + SF_MID_PAD; i += SF_THREADS) lds[SF_MIDB0 + i] = sf_dense3::bc();
  sf_dense3::acc_t acc3[SF_ACCS][SF_RJ][SF_VK];
#else
  for (int i = tid; i < SF_MID_SLOTS * SF_MID_STRIDE + SF_MID_PAD; i += SF_THREADS) lds[SF_MID0 + i] = sf_dense2::bc();
#endif
  sf_dense::acc_t acc1[SF_ACCS][SF_RJ][SF_VK];
  sf_dense2::acc_t acc2[SF_ACCS][SF_RJ][SF_VK];
  // (the arriving plane's values at the thread's own points, kept for the step that reads them again: codegen)
  sf_dense::acc_t carry1[SF_RJ][SF_VK];
  sf_dense2::acc_t carry2[SF_RJ][SF_VK];
#if SF_NST == 3
  sf_dense3::acc_t carry3[SF_RJ][SF_VK];
#endif
#pragma unroll
  for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
    for (int v = 0; v < SF_VK; ++v) {
      carry1[r][v] = (sf_dense::acc_t)0;
      carry2[r][v] = (sf_dense2::acc_t)0;
#if SF_NST == 3
      carry3[r][v] = (sf_dense3::acc_t)0;
#endif
    }
#pragma unroll
  for (int a = 0; a < SF_ACCS; ++a)
#pragma unroll
    for (int r = 0; r < SF_RJ; ++r)
#pragma unroll
      for (int v = 0; v < SF_VK; ++v) {
        acc1[a][r][v] = (sf_dense::acc_t)0;
        acc2[a][r][v] = (sf_dense2::acc_t)0;
#if SF_NST == 3
        acc3[a][r][v] = (sf_dense3::acc_t)0;
#endif
      }
#pragma unroll
  for (int a = 0; a < SF_AHEAD; ++a) sf_dma_plane(cx, p_begin + a, p_begin + a < p_end, a % SF_IN_SLOTS);
  // output plane q leaves at step q + SFD_DLAST + 1 + SFD2_DLAST (+ 1 + SFD3_DLAST); input planes up to ce + R - 1 are read
#if SF_NST == 3
  const int p_stop = cx.ce + SFD_DLAST + 1 + SFD2_DLAST + 1 + SFD3_DLAST;
#define SF_ACCS_ARGS acc1, acc2, carry1, carry2, acc3, carry3
#define SF_M2(n) ((mw2 + (n)) % SF_MID2_SLOTS)
#else
  const int p_stop = cx.ce + SFD_DLAST + 1 + SFD2_DLAST;
#define SF_ACCS_ARGS acc1, acc2, carry1, carry2
#define SF_M2(n) 0
#endif
  int slot = 0, mw = 0, mw2 = 0;  // the input plane's slot; the slots operator 1's and operator 2's planes go to
  for (int p = p_begin; p < p_stop; p += SF_ACCS) {
    sf_step_t2<0>(lds, out, sc, cx, p, p_begin, p_end, slot, mw, SF_M2(0), SF_ACCS_ARGS);
    sf_step_t2<1>(lds, out, sc, cx, p + 1, p_begin, p_end, (slot + 1) % SF_IN_SLOTS, (mw + 1) % SF_MID_SLOTS, SF_M2(1), SF_ACCS_ARGS);
    sf_step_t2<2>(lds, out, sc, cx, p + 2, p_begin, p_end, (slot + 2) % SF_IN_SLOTS, (mw + 2) % SF_MID_SLOTS, SF_M2(2), SF_ACCS_ARGS);
#if SF_ACCS == 5
    sf_step_t2<3>(lds, out, sc, cx, p + 3, p_begin, p_end, (slot + 3) % SF_IN_SLOTS, (mw + 3) % SF_MID_SLOTS, SF_M2(3), SF_ACCS_ARGS);
    sf_step_t2<4>(lds, out, sc, cx, p + 4, p_begin, p_end, (slot + 4) % SF_IN_SLOTS, (mw + 4) % SF_MID_SLOTS, SF_M2(4), SF_ACCS_ARGS);
#endif
    slot = (slot + SF_ACCS) % SF_IN_SLOTS;
    mw = (mw + SF_ACCS) % SF_MID_SLOTS;
#if SF_NST == 3
    mw2 = (mw2 + SF_ACCS) % SF_MID2_SLOTS;
#endif
  }
  (void)mw2;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (requests of planes nobody reads: landed before the LDS is given back)
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
#pragma unroll
  for (int a = 0; a < SF_AHEAD; ++a) sf_dma_plane(cx, p_begin + a, p_begin + a < p_end, a % SF_IN_SLOTS);
  int slot = 0;
  for (int p = p_begin; p < p_end; p += SF_ACCS) {
    sf_step_stream<0>(lds, out, sc, cx, p, p_begin, p_end, slot, acc);
    sf_step_stream<1>(lds, out, sc, cx, p + 1, p_begin, p_end, (slot + 1) % SF_IN_SLOTS, acc);
    sf_step_stream<2>(lds, out, sc, cx, p + 2, p_begin, p_end, (slot + 2) % SF_IN_SLOTS, acc);
    sf_step_stream<3>(lds, out, sc, cx, p + 3, p_begin, p_end, (slot + 3) % SF_IN_SLOTS, acc);
    sf_step_stream<4>(lds, out, sc, cx, p + 4, p_begin, p_end, (slot + 4) % SF_IN_SLOTS, acc);
#if SF_ACCS == 7
    sf_step_stream<5>(lds, out, sc, cx, p + 5, p_begin, p_end, (slot + 5) % SF_IN_SLOTS, acc);
    sf_step_stream<6>(lds, out, sc, cx, p + 6, p_begin, p_end, (slot + 6) % SF_IN_SLOTS, acc);
#endif
    slot = (slot + SF_ACCS) % SF_IN_SLOTS;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (requests of planes nobody reads: landed before the LDS is given back)
#else
  sf_pair regs[SF_NLOADS];
  sf_load_plane(cx, p_begin, regs, true);
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
