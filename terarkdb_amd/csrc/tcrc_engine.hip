// tcrc engine: batched CRC-32C kernels for gfx950 and the C ABI around them.
//
// Algorithm (all arithmetic over GF(2), register convention of gf2.h):
//   A buffer is walked by a group of G lanes.  One "row" is D = 16*G bytes; lane j
//   owns the 16-byte column j of every row and keeps four independent 32-bit
//   states, one per dword of its column.  Moving one row down is
//       s_i <- Z_D(s_i) ^ w_i          (Z_D = advance through D zero bytes)
//   and Z_D is four table lookups (one per state byte).  The tables live in LDS,
//   replicated 32 times with the replica index in the bank bits, so that lane l
//   only ever touches bank (l mod 32): every ds_read_b32 is conflict-free
//   whatever the data is.  Rows are aligned to D-byte ADDRESS boundaries, so every
//   lane load is an aligned 16-byte load and each wave instruction fetches whole
//   128-byte lines wherever a buffer starts.  The bytes of the first row before the
//   message are zeroed (leading zeros do not change a CRC remainder) and the
//   initial register is XORed into the first four message bytes; the bytes of the
//   last row after the message are zeroed too and taken back out at the end by one
//   multiplication with x^(-8 * their count).  After the last row the 4*G states
//   are reduced to one register with log2(G)+4 more (unreplicated) table steps and
//   wave shuffles.
//
// What it computes is the reference's crc32c::Extend (util/crc32c.cc:356-398,
// 1226-1228), Mask (util/crc32c.h:38-41), the SST trailer rule
// (table/block_based_table_builder.cc:598-601) and the verify comparisons of
// table/block_fetcher.cc:41-48,64 and db/log_reader.cc:388-390.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "tcrc_internal.h"
#include "gf2.h"

using namespace tcrc_internal;

namespace {

constexpr int kThreads = 1024;       // one workgroup per CU (LDS-bound), 16 waves
constexpr int kWavesPerWg = kThreads / 64;
constexpr int kMainDwords = 32768;   // 4 tables x 256 entries x 32 replicas
constexpr uint32_t kMaskDelta = 0xa282ead8u;
// Longest span one lane group walks (row arithmetic is 32-bit); longer inputs go through the long-buffer path.
constexpr uint32_t kMaxSpan = 1u << 30;

template <int G> struct Geo;
template <> struct Geo<4>  { static constexpr int kLog2 = 2; };
template <> struct Geo<8>  { static constexpr int kLog2 = 3; };
template <> struct Geo<16> { static constexpr int kLog2 = 4; };
template <> struct Geo<64> { static constexpr int kLog2 = 6; };


typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef v4u v4u_any __attribute__((aligned(1)));

// 16 bytes from any address: one global_load_dwordx4 (gfx950 handles unaligned vector loads).
// NT marks the line as streamed-once.
// The pointer is cast to the global address space: a generic pointer makes the compiler emit flat loads, which
// also occupy the LDS queue and force every wait to be a full one.
typedef const v4u_any __attribute__((address_space(1)))* gptr16;
template <bool NT>
__device__ __forceinline__ uint4 load16(const uint8_t* p) {
  v4u v;
  if (NT) {
    v = __builtin_nontemporal_load((gptr16)p);
  } else {
    v = *(gptr16)p;
  }
  return make_uint4(v.x, v.y, v.z, v.w);
}

// 16 bytes to any address: one global_store_dwordx4.
typedef v4u_any __attribute__((address_space(1)))* gptr16w;
__device__ __forceinline__ void store16(uint8_t* p, const uint4& v) {
  v4u x;
  x.x = v.x;
  x.y = v.y;
  x.z = v.z;
  x.w = v.w;
  // (default policy on purpose: marked as streamed-once, these stores - partial lines when the rows follow the
  // source's alignment - made the framing kernel twice as slow, and 12 % slower with rows on the image's lines)
  *(gptr16w)p = x;
}

// Z_D via the replicated tables.  LDS layout of the main set (128 KiB): entry b of the pair (T0, T1) is the 256
// bytes at b * 256 - 32 replicas of T0[b], then 32 replicas of T1[b] - and the pair (T2, T3) follows at 64 KiB
// in the same form.  The byte address of a lookup is then (state byte << 8) | (lane mod 32) * 4 | region bit 16,
// which ONE byte-permute instruction forms from the state and a per-lane constant (the odd table of a pair is
// the instruction's immediate offset 128); lane l still only touches bank l mod 32.
// LB = (lane mod 32) * 4 | 0x10000.  v_perm_b32: result byte i = selector byte i picks 0-3 from LB, 4-7 from
// c, 0x0c = zero.
__device__ __forceinline__ uint32_t fold_main(const uint32_t* lds, uint32_t LB, uint32_t c) {
  const char* base = (const char*)lds;
  const uint32_t a0 = __builtin_amdgcn_perm(c, LB, 0x0c0c0400u);
  const uint32_t a1 = __builtin_amdgcn_perm(c, LB, 0x0c0c0500u);
  const uint32_t a2 = __builtin_amdgcn_perm(c, LB, 0x0c020600u);
  const uint32_t a3 = __builtin_amdgcn_perm(c, LB, 0x0c020700u);
  return *(const uint32_t*)(base + a0) ^ *(const uint32_t*)(base + a1 + 128) ^ *(const uint32_t*)(base + a2) ^
         *(const uint32_t*)(base + a3 + 128);
}

// Z_n via one unreplicated 4 KiB set.
__device__ __forceinline__ uint32_t fold_aux(const uint32_t* t, uint32_t c) {
  return t[c & 0xffu] ^ t[256u + ((c >> 8) & 0xffu)] ^ t[512u + ((c >> 16) & 0xffu)] ^ t[768u + (c >> 24)];
}

// The remainder of one span per lane group: bytes [ptr, ptr + len) fed to the register ~init.  Returns the
// register BEFORE the final inversion (and before any suffix byte), meaningful on the last lane of every
// group whose `valid` is set; groups that are not valid walk nothing.  All 64 lanes of the wave must call it
// together (the row count is agreed on with shuffles).  safe: 4 KiB of readable global memory that lane
// groups without rows load from where the others load rows.
// reduction sets, then x^(-8t) * x^(32 l / UL) for t < D, l < UL
template <int G> struct AuxDwords {
  static constexpr int value = (1 + Geo<G>::kLog2) * 1024 + 16 * G * (G <= 16 ? 4 : 1);
};

// COPY: the bytes are also written to [dstp, dstp + len) as they pass through the registers (log framing: the
// payload is read once, copied and checksummed); lane groups without rows store to `sink` (4 KiB of writable
// scratch) where the others store rows.  PHASE_DST: the rows are aligned to the DESTINATION's 128-byte lines
// instead of the source's, so that the stores are the aligned ones and the loads take whatever alignment is left.
template <int G, bool NT, bool COPY = false, bool PHASE_DST = false>
__device__ __forceinline__ uint32_t span_remainder(const uint32_t* lds, uint32_t LB, int lane, const uint8_t* ptr,
                                                   uint32_t len, uint32_t init, bool valid, const uint8_t* safe,
                                                   uint8_t* dstp = nullptr, uint8_t* sink = nullptr) {
  constexpr int D = 16 * G;
  constexpr int LOG2G = Geo<G>::kLog2;
  // lanes of a group that share the un-shift multiplication (a whole wave per buffer keeps it on one lane:
  // its table would not fit next to the others, and its buffers are long enough not to care)
  constexpr int UL = G <= 16 ? 4 : 1;
  const int j = lane & (G - 1);
  const uint32_t* aux = lds + kMainDwords;
  const uint32_t r0 = ~init;
  const bool live = valid && len >= 16;  // shorter messages take the byte path at the end
  const uint32_t elen = live ? len : 0u;
    // Rows are aligned to D-byte ADDRESS boundaries, so every 16-byte load is aligned and every
  // row of a buffer is whole cache lines.  The bytes of the first row before the message and of
  // the last row after it are read (same lines as message bytes) and masked to zero: leading
  // zeros leave a CRC remainder unchanged, the `tpad` trailing ones are undone after the
  // reduction by one multiplication with x^(-8 tpad).
  const uintptr_t phase_of = (COPY && PHASE_DST) ? (uintptr_t)dstp : (uintptr_t)ptr;
  const uint32_t a0 = live ? (uint32_t)(phase_of & (uintptr_t)(D - 1)) : 0u;
  const uint32_t span = a0 + elen;
  const uint32_t R = (span + D - 1) / D;
  const uint32_t tpad = R * D - span;
  uint32_t Rmax = R;
#pragma unroll
  for (int o = 32; o >= G; o >>= 1) {
    const uint32_t other = __shfl_xor(Rmax, o);
    Rmax = other > Rmax ? other : Rmax;
  }
  // the same in every lane now: kept in a scalar register, so that the row loops and the row classes
  // below cost scalar instructions
  Rmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Rmax);
  // A buffer with fewer rows than the longest of its wave starts later, so that all end together.
  // message offset of this lane's column in row r is m0 + r*D
  const int m0 = 16 * j - (int)a0 - (int)((Rmax - R) * D);
  uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;

  auto fold_plain = [&](const uint4& v) {
    s0 = fold_main(lds, LB, s0) ^ v.x;
    s1 = fold_main(lds, LB, s1) ^ v.y;
    s2 = fold_main(lds, LB, s2) ^ v.z;
    s3 = fold_main(lds, LB, s3) ^ v.w;
  };
  // A column that may stick out of the message at either end or hold some of its first four
  // bytes (which carry the initial register): loaded only if it overlaps the message ...
  auto load_checked = [&](uint32_t r) -> uint4 {
    const int m = m0 + (int)(r * D);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    // default cache policy, not the streaming one of the whole rows: the line a span starts or ends in is
    // usually also the line its neighbour ends or starts in, and is wanted again within microseconds
    // (4 KiB blocks packed end to end: 6.7 TB/s against 6.2 with streaming loads here too)
    if constexpr (COPY && PHASE_DST) {
      // Rows follow the destination's lines here, so a column that sticks out of the message would be read from
      // bytes that are not in the message's lines at all (before the source allocation, or past its end).  Such
      // a column is read from the nearest sixteen bytes INSIDE the message instead and shifted into place; what
      // is shifted in is zero, which is what the trimming makes of those bytes anyway.
      if (live && m > -16 && m < (int)elen) {
        const int mc = m < 0 ? 0 : (m + 16 > (int)elen ? (int)elen - 16 : m);
        v = load16<false>(ptr + mc);
        const int sh = mc - m;  // > 0: column byte k is loaded byte k - sh; < 0: loaded byte k + |sh|
        if (sh != 0) {
          unsigned long long lo = ((unsigned long long)v.y << 32) | v.x;
          unsigned long long hi = ((unsigned long long)v.w << 32) | v.z;
          if (sh > 0) {
            const uint32_t b = 8u * (uint32_t)sh;
            if (b >= 64u) {
              hi = lo << (b - 64u);
              lo = 0;
            } else {
              hi = (hi << b) | (lo >> (64u - b));
              lo <<= b;
            }
          } else {
            const uint32_t b = 8u * (uint32_t)(-sh);
            if (b >= 64u) {
              lo = hi >> (b - 64u);
              hi = 0;
            } else {
              lo = (lo >> b) | (hi << (64u - b));
              hi >>= b;
            }
          }
          v = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        }
      }
      return v;
    }
    if (live && m > -16 && m < (int)elen) v = load16<false>(ptr + m);
    return v;
  };
  // ... and trimmed to the message, with the initial register folded into message bytes 0..3.
  auto fold_checked = [&](uint4 v, uint32_t r) {
    const int m = m0 + (int)(r * D);
    if constexpr (COPY) {  // a column that lies wholly inside the message goes out as it came in
      if (live && m >= 0 && m + 16 <= (int)elen) store16(dstp + m, v);
    }
    if (live && m > -16 && m < (int)elen && (m < 4 || m + 16 > (int)elen)) {
      unsigned long long lo = ((unsigned long long)v.y << 32) | v.x;
      unsigned long long hi = ((unsigned long long)v.w << 32) | v.z;
      if (m < 0) {  // bytes [0, -m) precede the message
        const uint32_t k = (uint32_t)(-m);
        if (k >= 8u) {
          lo = 0;
          hi &= ~0ull << (8u * (k - 8u));
        } else {
          lo &= ~0ull << (8u * k);
        }
      }
      if (m + 16 > (int)elen) {  // only the first `keep` bytes belong to the message
        const uint32_t keep = elen - (uint32_t)m;
        if (keep <= 8u) {
          hi = 0;
          if (keep < 8u) lo &= (1ull << (8u * keep)) - 1ull;
        } else {
          hi &= (1ull << (8u * (keep - 8u))) - 1ull;
        }
      }
      if (m < 4) {  // message bytes 0..3 sit at column bytes -m .. -m+3
        if (m >= 0) {
          lo ^= (unsigned long long)(r0 >> (8 * m));
        } else {
          const uint32_t k = (uint32_t)(-m);
          if (k < 8u) {
            lo ^= (unsigned long long)r0 << (8u * k);
            if (k > 4u) hi ^= (unsigned long long)(r0 >> (64u - 8u * k));
          } else {
            hi ^= (unsigned long long)r0 << (8u * (k - 8u));
          }
        }
      }
      v = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    }
    fold_plain(v);
  };
  // a row that is whole for every buffer of the wave
  auto fold_whole = [&](const uint4& v, uint32_t r) {
    if constexpr (COPY) {
      if (live) store16(dstp + (m0 + (int)(r * D)), v);
    }
    fold_plain(v);
  };
  // Rows go in bursts of four: four 16-byte loads per lane are issued back to back (512
  // contiguous bytes per buffer), then folded in order as they arrive.  Bursts without overlap
  // inside a wave measured faster than keeping loads in flight across the folds; the other
  // waves of the CU cover the latency.  The Rmax mod 4 odd rows come first, one burst.
  uint32_t r = 0;
  // The common shape - every buffer of the wave starts and ends on a row boundary and has the
  // same number of rows, a multiple of four - needs no per-group checks at all: only lane 0 of
  // each group folds the initial register into its first dword.
  if ((Rmax & 3u) == 0u && __all(live && a0 == 0u && tpad == 0u && R == Rmax)) {
    const uint8_t* q = ptr + 16 * j;
    {
      uint4 a = load16<NT>(q);
      const uint4 b = load16<NT>(q + D);
      const uint4 c = load16<NT>(q + 2 * D);
      const uint4 d = load16<NT>(q + 3 * D);
      if constexpr (COPY) {
        uint8_t* w = dstp + 16 * j;
        store16(w, a);
        store16(w + D, b);
        store16(w + 2 * D, c);
        store16(w + 3 * D, d);
      }
      if (j == 0) a.x ^= r0;
      fold_plain(a);
      fold_plain(b);
      fold_plain(c);
      fold_plain(d);
    }
    for (r = 4u; r < Rmax; r += 4u) {
      q += 4 * D;
      const uint4 a = load16<NT>(q);
      const uint4 b = load16<NT>(q + D);
      const uint4 c = load16<NT>(q + 2 * D);
      const uint4 d = load16<NT>(q + 3 * D);
      if constexpr (COPY) {
        uint8_t* w = dstp + 16 * j + (size_t)r * D;
        store16(w, a);
        store16(w + D, b);
        store16(w + 2 * D, c);
        store16(w + 3 * D, d);
      }
      fold_plain(a);
      fold_plain(b);
      fold_plain(c);
      fold_plain(d);
    }
  }
  if (r == 0u && Rmax) {
    // General shape.  For one buffer only its first row (leading bytes outside the message, and the
    // initial register in message bytes 0..3) and its last row (trailing bytes outside) need the
    // trimming of fold_checked; buffers end together, so the last row is row Rmax-1 for all of them,
    // and a buffer's first row is row Rmax-R.  Rows [head_end, tail_begin) are whole for every buffer
    // of the wave and are folded without trimming.
    // (a buffer that starts in the last three bytes of a row has part of message bytes 0..3, which
    // carry the initial register, in its second row too)
    uint32_t fmax = live ? Rmax - R + (a0 > (uint32_t)(D - 4) ? 1u : 0u) : 0u;
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) {
      const uint32_t other = __shfl_xor(fmax, o);
      fmax = other > fmax ? other : fmax;
    }
    fmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)fmax);
    const uint32_t head_end = fmax + 1u < Rmax ? fmax + 1u : Rmax;
    const uint32_t tail_begin = Rmax - 1u > head_end ? Rmax - 1u : head_end;
    // Checked bursts: up to four row loads are issued together - a load is skipped by lanes whose column lies
    // outside the message - and then folded in order, trimmed or plain by the row's class.
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    auto bursts = [&](uint32_t from, uint32_t to) {
      for (uint32_t x = from; x < to; x += 4u) {
        const uint32_t cnt = to - x;  // rows in this burst if fewer than four
        const uint4 a = load_checked(x);
        const uint4 b = cnt > 1u ? load_checked(x + 1u) : z;
        const uint4 c = cnt > 2u ? load_checked(x + 2u) : z;
        const uint4 d = cnt > 3u ? load_checked(x + 3u) : z;
        if (x >= head_end && x < tail_begin) fold_whole(a, x); else fold_checked(a, x);
        if (cnt > 1u) {
          if (x + 1u >= head_end && x + 1u < tail_begin) fold_whole(b, x + 1u); else fold_checked(b, x + 1u);
        }
        if (cnt > 2u) {
          if (x + 2u >= head_end && x + 2u < tail_begin) fold_whole(c, x + 2u); else fold_checked(c, x + 2u);
        }
        if (cnt > 3u) {
          if (x + 3u >= head_end && x + 3u < tail_begin) fold_whole(d, x + 3u); else fold_checked(d, x + 3u);
        }
      }
    };
    // The rows before head_end go in checked bursts.  Then the whole rows, as many of them as make a multiple
    // of four, go through a window of four loads in flight: a row is folded and its register is at once given
    // to the load of the row four further on, so the memory latency is paid once and not once per four rows
    // (buffers of a few KiB are bound by that latency, not by bandwidth).  That loop has no branch in it -
    // lane groups that walk nothing read a harmless address instead of being masked off - because only then
    // does the compiler count the loads in flight instead of waiting for all of them.
    const uint32_t mid_end = head_end + ((tail_begin - head_end) & ~3u);
    bursts(0u, head_end);
    if (mid_end > head_end) {
      const uint8_t* q = live ? ptr + (m0 + (int)(head_end * D)) : safe;
      const uint32_t qstep = live ? 4u * D : 0u;
      uint4 a = load16<NT>(q);
      uint4 b = load16<NT>(q + D);
      uint4 c = load16<NT>(q + 2 * D);
      uint4 d = load16<NT>(q + 3 * D);
      // COPY: where the window's rows go (lane groups that walk nothing write to the sink, branch-free like the loads)
      uint8_t* w = nullptr;
      uint32_t wstep = 0;
      if constexpr (COPY) {
        w = live ? dstp + (m0 + (int)(head_end * D)) : sink + 16 * lane;
        wstep = qstep;
      }
      for (uint32_t x = head_end + 4u; x < mid_end; x += 4u) {
        q += qstep;
        if constexpr (COPY) store16(w, a);
        fold_plain(a);
        a = load16<NT>(q);
        if constexpr (COPY) store16(w + D, b);
        fold_plain(b);
        b = load16<NT>(q + D);
        if constexpr (COPY) store16(w + 2 * D, c);
        fold_plain(c);
        c = load16<NT>(q + 2 * D);
        if constexpr (COPY) store16(w + 3 * D, d);
        fold_plain(d);
        d = load16<NT>(q + 3 * D);
        if constexpr (COPY) w += wstep;
      }
      if constexpr (COPY) {
        store16(w, a);
        store16(w + D, b);
        store16(w + 2 * D, c);
        store16(w + 3 * D, d);
      }
      fold_plain(a);
      fold_plain(b);
      fold_plain(c);
      fold_plain(d);
    }
    // What is left is one more burst: the whole rows that did not make a multiple of four (up to three) - as whole
    // as the others, so plain streaming loads counted with scalar instructions - and behind them, in flight
    // together with them, the last row, which alone goes through the checks (a checked row costs some fifty
    // vector instructions on top of its thirty-two).
    if (tail_begin + 1u == Rmax) {
      const uint32_t left = tail_begin - mid_end;
      const uint8_t* q = live ? ptr + (m0 + (int)(mid_end * D)) : safe;
      const uint32_t qd = live ? (uint32_t)D : 0u;
      const uint4 a = left > 0u ? load16<NT>(q) : z;
      const uint4 b = left > 1u ? load16<NT>(q + qd) : z;
      const uint4 c = left > 2u ? load16<NT>(q + 2u * qd) : z;
      const uint4 d = load_checked(tail_begin);
      if (left > 0u) fold_whole(a, mid_end);
      if (left > 1u) fold_whole(b, mid_end + 1u);
      if (left > 2u) fold_whole(c, mid_end + 2u);
      fold_checked(d, tail_begin);
    }
  }

  if constexpr (COPY) {
    // The columns that stick out of the message at its two ends were not stored: the message's first and last
    // sixteen bytes go out once more, whole (they overlap what the inner columns wrote with the same bytes).
    if (live && j < 2) {
      const uint32_t at = j == 0 ? 0u : elen - 16u;
      store16(dstp + at, load16<false>(ptr + at));
    }
  }
  // Column states -> one value per lane (bytes j*16 .. j*16+15 of the last row).
  uint32_t u = fold_aux(aux, s0) ^ s1;
  u = fold_aux(aux, u) ^ s2;
  u = fold_aux(aux, u) ^ s3;
  // Lanes -> lane G-1: at step d lanes [G-2d, G-d) hand Z_{16d}(u) to [G-d, G).
#pragma unroll
  for (int st = LOG2G; st >= 1; --st) {
    const int dist = 1 << (st - 1);
    const uint32_t t = fold_aux(aux + st * 1024, u);
    u ^= __shfl_up(t, dist, G);
  }

  // The last dword still owes its own four bytes; then the trailing zeros are taken back out:
  // r * x^(-8 tpad).  With UL lanes the 32 shift-and-xor steps are split: lane l takes the bits
  // [32 l / UL, 32 (l+1) / UL) of r (counted from x^0) against x^(-8 tpad) * x^(32 l / UL) from the table.
  uint32_t rfin = fold_aux(aux, u);  // meaningful on the group's last lane
  if (UL == 1) {
    if (live && tpad && j == G - 1) rfin = gf2_mulmod_dev(rfin, aux[(1 + LOG2G) * 1024 + tpad]);
  } else if (__any(live && tpad != 0u)) {
    const uint32_t rr = __shfl(rfin, lane | (G - 1));
    const int l = j & (UL - 1);
    uint32_t b = aux[(1 + LOG2G) * 1024 + tpad * UL + l];
    uint32_t part = 0;
#pragma unroll
    for (int k = 0; k < 32 / UL; ++k) {
      part ^= ((rr >> (31 - (32 / UL) * l - k)) & 1u) ? b : 0u;
      b = (b >> 1) ^ ((b & 1u) ? 0x82f63b78u : 0u);
    }
#pragma unroll
    for (int o = 1; o < UL; o <<= 1) part ^= __shfl_xor(part, o);
    if (tpad) rfin = part;
  }
  uint32_t reg = rfin;
  if (valid && !live && j == G - 1) {
    reg = r0;
    const uint32_t* T0 = aux + 768;  // one-byte step table
    for (uint32_t k = 0; k < len; ++k) {
      const uint8_t byte = ptr[k];
      if constexpr (COPY) dstp[k] = byte;
      reg = T0[(reg ^ byte) & 0xffu] ^ (reg >> 8);
    }
  }
  return reg;
}

// where the second kernel argument starts in the argument block
constexpr size_t kLongArgsOffset = (sizeof(BatchParams) + alignof(LongArgs) - 1) / alignof(LongArgs) * alignof(LongArgs);
static_assert(sizeof(tcrc_desc) == 16 && offsetof(LongArgs, inl) % 16 == 0 && kLongArgsOffset % 16 == 0,
              "descriptors in the kernel arguments are fetched as 16-byte words");

// The main lookup set goes from its 4 KiB in global memory into its 32 replicas in LDS.  Dword i of the set:
// pair i / 16384, entry (i / 64) mod 256, table of the pair (i / 32) mod 2, replica i mod 32 - so four
// consecutive dwords hold one value: a thread fetches it once and writes sixteen bytes.
__device__ __forceinline__ void fill_main_tables(uint32_t* lds, const uint32_t* tab_main, int tid) {
  static_assert(kMainDwords % (4 * kThreads) == 0, "whole passes");
#pragma unroll
  for (int q = tid; q < kMainDwords / 4; q += kThreads) {
    const int i = q * 4;
    const uint32_t v = tab_main[((i >> 14) * 2 + ((i >> 5) & 1)) * 256 + ((i >> 6) & 255)];
    reinterpret_cast<uint4*>(lds)[q] = make_uint4(v, v, v, v);
  }
}

// What a batch launch does with its results besides writing them to arrays.
constexpr int kModePlain = 0;     // nothing
constexpr int kModeSegments = 1;  // the items are segments of long buffers: results are shifted and folded per buffer
constexpr int kModeInImage = 2;   // results are also stored into the image (TCRC_F_STORE_AFTER, kFlagStampBefore)
template <int MODE> struct SegArgs { typedef NoLongArgs type; };
template <> struct SegArgs<kModeSegments> { typedef LongArgs type; };

// One segment's remainder enters its buffer's accumulator line: XOR, then the arrival count; whoever
// completes the count takes the value out, leaves the line zeroed for the next launch and writes the result.
// Device-scope atomics are performed at the memory side, past the per-XCD L2s, so they need no cache
// write-back or invalidation (which a __threadfence() would do, for every wave, to the whole L2); all the
// protocol needs is that a wave's XOR has been performed before its count is added, i.e. that the XOR's
// acknowledgement has come back: a wait on the vector-memory counter.
__device__ __forceinline__ void seg_arrive(uint32_t* line, uint32_t value, uint32_t count, uint32_t total,
                                           uint32_t* out, const DoneSignal& done, uint32_t nbuf) {
  (void)__hip_atomic_fetch_xor(line, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t before = __hip_atomic_fetch_add(line + 1, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (before + count == total) {  // every other segment's XOR was acknowledged before its count was added
    const uint32_t r = __hip_atomic_exchange(line, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_exchange(line + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!done.flag) {
      *out = r;
    } else {
      // a host thread is watching for the flag: the result is written through to memory and counted off
      // behind a system-scope release (a store is acknowledged to its wave by the L2, which may still hold
      // it; and the flag takes another channel of the L2)
      __hip_atomic_store(out, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      bool last = nbuf == 1u;
      if (!last) {
        last = __hip_atomic_fetch_add(done.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == nbuf;
        if (last) (void)__hip_atomic_exchange(done.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (last) __hip_atomic_store(done.flag, done.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// kModeSegments: the items are segments of long buffers described by `la` (tcrc_long.hip); otherwise `la` is
// empty.  The modes are separate instantiations so that what one needs (registers, above all) costs the others
// nothing: the plain one is the benchmark's.
template <int G, bool NT, int MODE>
__global__ __launch_bounds__(kThreads) void k_crc_batch(BatchParams p, typename SegArgs<MODE>::type la) {
  constexpr bool SEG = MODE == kModeSegments;
  constexpr int GPW = 64 / G;  // groups (buffers) per wave
  constexpr int kAuxDwords = AuxDwords<G>::value;
  __shared__ __align__(16) uint32_t lds[kMainDwords + kAuxDwords];
  // SEG: segments before every buffer of the launch, and the descriptors that came in the kernel arguments
  __shared__ uint32_t seg_first_lds[SEG ? kLongLdsBuffers + 1 : 1];
  __shared__ uint4 seg_inl_lds[SEG ? kLongInline : 1];

  const int tid = threadIdx.x;
  if (p.n_dev && *p.n_dev == 0) return;  // an empty list (uniform over the grid): nothing to set up
  fill_main_tables(lds, p.tab_main, tid);
  static_assert(kAuxDwords % 4 == 0, "the small sets are copied sixteen bytes at a time");
  for (int q = tid; q < kAuxDwords / 4; q += kThreads)
    reinterpret_cast<uint4*>(lds + kMainDwords)[q] = reinterpret_cast<const uint4*>(p.tab_aux)[q];
  if constexpr (SEG) {
    if (la.seg_first) {
      for (uint32_t i = tid; i <= la.nbuf; i += kThreads) seg_first_lds[i] = la.seg_first[i] - la.first_base;
    } else {
      // Descriptors that came in the kernel arguments go to LDS, where they can be indexed.  The argument
      // block is ordinary memory: thread i fetches descriptor i from it with vector loads (walking the
      // arguments with constant indices on one thread instead costs about 2 us for 64 buffers, with every
      // other wave waiting at the barrier below).
      const uint8_t* args = (const uint8_t*)__builtin_amdgcn_kernarg_segment_ptr() + kLongArgsOffset;
      if ((uint32_t)tid < la.nbuf) {
        seg_inl_lds[tid] = *(const uint4*)(args + offsetof(LongArgs, inl) + (size_t)tid * sizeof(tcrc_desc));
        seg_first_lds[tid] = *(const uint32_t*)(args + offsetof(LongArgs, inl_first) + (size_t)tid * 4u);
      }
      if (tid == 0) seg_first_lds[la.nbuf] = (uint32_t)p.n;  // all segments of the launch
    }
  }
  __syncthreads();

  const int lane = tid & 63;
  const int j = lane & (G - 1);
  const int grp = lane / G;
  const uint32_t LB = (uint32_t)(lane & 31) * 4u | 0x10000u;
  const uint32_t* aux = lds + kMainDwords;
  // rounds are dealt to workgroups first, then to the waves of a workgroup, so that a small
  // batch still spreads over all CUs
  const uint64_t wave = (uint64_t)blockIdx.x + (uint64_t)gridDim.x * (tid >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerWg;
  uint64_t n = p.n;
  if (p.n_dev) {
    const uint64_t nd = *p.n_dev;
    n = nd < n ? nd : n;
  }
  const uint64_t nrounds = (n + GPW - 1) / GPW;
  // SEG: what this wave has gathered for one buffer and not yet delivered
  uint32_t pend_owner = 0, pend_total = 0, pend_value = 0, pend_count = 0;

  // the descriptor of the next round is fetched while the current round runs
  tcrc_desc d_next = {0, 0, 0};
  if (p.desc && wave < nrounds && wave * GPW + grp < n) d_next = p.desc[wave * GPW + grp];

  // Which rounds this wave takes.  Independent buffers: round wave, wave + nwaves, ... (dealt to workgroups
  // first, then to the waves of a workgroup, so that a small batch still spreads over all CUs).  Segments of
  // long buffers: a run of consecutive rounds, and consecutive runs for the waves of a workgroup - what a
  // wave, and then a workgroup, gathers is then mostly for one buffer and reaches that buffer's accumulator
  // line as one arrival.
  uint64_t round_first = wave, round_last = nrounds, round_step = nwaves;
  if constexpr (SEG) {
    const uint64_t per_wg = (nrounds + gridDim.x - 1) / gridDim.x;
    const uint64_t wpw = per_wg < (uint64_t)kWavesPerWg ? per_wg : (uint64_t)kWavesPerWg;  // active waves per workgroup
    const uint64_t k = (nrounds + (uint64_t)gridDim.x * wpw - 1) / ((uint64_t)gridDim.x * wpw);
    const uint64_t w = (uint64_t)blockIdx.x * wpw + (uint64_t)(tid >> 6);
    round_step = 1;
    round_first = (uint64_t)(tid >> 6) < wpw ? w * k : nrounds;
    round_last = round_first + k < nrounds ? round_first + k : nrounds;
  }
  // Results stored into the image (a trailer behind its block, TCRC_F_STORE_AFTER; a checksum into the record
  // header in front of its span) are kept back: lane i of a group holds the result of the group's i-th round
  // since the last flush, and after G rounds - or at the end - every lane stores one of them.  Five-byte stores
  // between the row loads of the next round cost 4 KiB blocks a tenth of their rate (5.9 TB/s against 6.7 for
  // results that go to an array); in bursts of G rounds, behind the wave's last load, a third of that (6.4 TB/s).
  constexpr bool defer = MODE == kModeInImage;
  uint32_t pend_result = 0, pend_rounds = 0;
  uint64_t pend_first = 0;
  auto flush_trailers = [&]() {
    const uint64_t buf = (pend_first + (uint64_t)j * round_step) * GPW + grp;
    if ((uint32_t)j < pend_rounds && buf < n) {
      uint64_t off = buf * p.stride;
      uint32_t len = p.block_len;
      if (p.desc) {
        const tcrc_desc dsc = p.desc[buf];
        off = dsc.offset;
        len = dsc.len > kMaxSpan ? 0u : dsc.len;
      }
      if (p.flags & TCRC_F_STORE_AFTER) {
        int sfx = p.uniform_suffix;
        if (p.suffix) sfx = p.suffix[buf];
        uint8_t* t = const_cast<uint8_t*>(p.base) + off + len;
        if (sfx >= 0) *t++ = (uint8_t)sfx;
        __builtin_memcpy(t, &pend_result, 4);
      }
      if ((p.flags & kFlagStampBefore) && len != 0u) {
        // into the record header of a log image being framed (the image is writable there)
        uint8_t* h = const_cast<uint8_t*>(p.base) + off - 6;
        __builtin_memcpy(h, &pend_result, 4);
      }
    }
    pend_rounds = 0;
  };
  for (uint64_t round = round_first; round < round_last; round += round_step) {
    const uint64_t buf = round * GPW + grp;
    const bool valid = buf < n;
    uint64_t off = 0;
    uint32_t len = 0, init = 0;
    bool oversize = false;
    const tcrc_desc dsc = d_next;
    if (p.desc) {
      const uint64_t nb = (round + nwaves) * GPW + grp;
      if (round + nwaves < nrounds && nb < n) d_next = p.desc[nb];
    }
    uint32_t seg_owner = 0, seg_after = 0, seg_total = 0, seg_pow_lo = 0, seg_pow_hi = 0;
    if constexpr (SEG) {
      if (valid) {
        // buffer that owns segment `buf`: the last b with seg_first[b] <= buf
        const uint32_t sidx = (uint32_t)buf;
        uint32_t lo = 0, hi = la.nbuf;
        while (hi - lo > 1u) {
          const uint32_t mid = (lo + hi) >> 1;
          if (seg_first_lds[mid] <= sidx) lo = mid; else hi = mid;
        }
        const uint32_t f0 = seg_first_lds[lo];
        tcrc_desc bd;
        if (la.bufs) {
          bd = la.bufs[lo];
        } else {
          const uint4 w = seg_inl_lds[lo];
          bd.offset = (uint64_t)w.x | ((uint64_t)w.y << 32);
          bd.len = w.z;
          bd.init = w.w;
        }
        seg_owner = lo;
        seg_total = seg_first_lds[lo + 1] - f0;
        seg_after = seg_total - 1u - (sidx - f0);  // whole segments that follow this one in its buffer
        const uint64_t end = (uint64_t)bd.len - ((uint64_t)seg_after << la.seg_shift);
        const uint64_t seg_len = 1ull << la.seg_shift;
        const uint64_t start = end > seg_len ? end - seg_len : 0ull;
        off = bd.offset + start;
        len = (uint32_t)(end - start);
        init = sidx == f0 ? bd.init : 0xffffffffu;  // later segments start from a zero register
        // the powers this segment's remainder will be multiplied by are fetched now, behind the row loop
        if (j == G - 1) {
          seg_pow_lo = la.pow_lo[seg_after & (kPowEntries - 1)];
          if (seg_after >= (uint32_t)kPowEntries) seg_pow_hi = la.pow_hi[seg_after >> 11];
        }
      }
    } else if (valid) {
      if (p.desc) {
        off = dsc.offset;
        len = dsc.len;
        init = dsc.init;
        if (len > kMaxSpan) {  // not walked: nothing is read, a verify counts it as a mismatch
          len = 0;
          oversize = true;
        }
      } else {
        off = buf * p.stride;
        len = p.block_len;
        init = p.uniform_init;
      }
    }
    // the stored value is fetched now so its latency hides behind the row loop
    uint32_t stored = 0;
    if (p.verify && valid && j == G - 1) {
      if (p.expected) {
        stored = p.expected[buf];
      } else {
        // behind the buffer - and behind its suffix byte when there is one, where a stamp with
        // TCRC_F_STORE_AFTER put it - or in front of it
        const uint32_t sfx_bytes = (p.uniform_suffix >= 0 || p.suffix) ? 1u : 0u;
        const uint8_t* e = (p.flags & TCRC_F_EXPECT_AFTER) ? p.base + off + len + sfx_bytes : p.base + off - 6;
        __builtin_memcpy(&stored, e, 4);  // one unaligned dword load
      }
    }
    const uint8_t* ptr = p.base + off;
    const uint32_t core = span_remainder<G, NT>(lds, LB, lane, ptr, len, init, valid, (const uint8_t*)p.tab_main);

    uint32_t seg_value = 0, round_result = 0;
    bool seg_have = false;
    if (valid && j == G - 1) {
      uint32_t reg = core;
      int sfx = p.uniform_suffix;
      if (p.suffix) sfx = p.suffix[buf];
      if (sfx >= 0) reg = aux[768u + ((reg ^ (uint32_t)sfx) & 0xffu)] ^ (reg >> 8);
      const uint32_t crc = ~reg;
      const uint32_t masked = ((crc >> 15) | (crc << 17)) + kMaskDelta;
      if constexpr (SEG) {
        // reg is the remainder of this segment alone; q whole segments follow it in its buffer, so it
        // enters the buffer's remainder advanced through q * segment_len bytes: x^(8 S q) is the product of
        // one entry of each level of the power table.  The final inversion rides on the last segment (q == 0).
        const uint32_t q = seg_after;
        uint32_t prod = gf2_mulmod_dev(reg, seg_pow_lo);
        if (q >= (uint32_t)kPowEntries) prod = gf2_mulmod_dev(prod, seg_pow_hi);
        seg_value = q == 0 ? ~prod : prod;
        seg_have = true;
      } else {
        const uint32_t result = (p.flags & TCRC_F_MASK) ? masked : crc;
        if (p.out) p.out[buf] = result;
        round_result = result;
      }
      if (p.verify) {
        const bool good = stored == masked && !oversize;
        if (p.ok) p.ok[buf] = good ? 1 : 0;
        if (!good) {
          atomicAdd(&p.counts[TCRC_CNT_MISMATCH], 1ull);
          atomicMin(&p.counts[TCRC_CNT_FIRST_BAD], (unsigned long long)buf);
        }
      }
    }
    if (defer) {
      const uint32_t r = __shfl(round_result, lane | (G - 1));
      if ((uint32_t)j == pend_rounds) pend_result = r;
      if (pend_rounds == 0u) pend_first = round;
      if (++pend_rounds == (uint32_t)G) flush_trailers();
    }
    if constexpr (SEG) {
      // The segments of one wave usually belong to one buffer: fold them with shuffles and touch the
      // accumulator once.  Accumulators sit one per 128-byte line so that different buffers do not queue
      // on the same line.
      const uint32_t first_owner = __shfl(seg_owner, G - 1);
      const uint32_t first_total = __shfl(seg_total, G - 1);
      const unsigned long long have = __ballot(seg_have);
      if (__all(!seg_have || seg_owner == first_owner)) {
        uint32_t v = seg_have ? seg_value : 0u;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v ^= __shfl_xor(v, o);
        if (have) {
          // kept until the wave meets another buffer, or the launch ends
          if (pend_count && pend_owner != first_owner) {
            if (lane == 0)
              seg_arrive(la.acc + (uint64_t)pend_owner * 32u, pend_value, pend_count, pend_total, p.out + pend_owner,
                         p.done, la.nbuf);
            pend_value = 0;
            pend_count = 0;
          }
          pend_owner = first_owner;
          pend_total = first_total;
          pend_value ^= v;
          pend_count += (uint32_t)__popcll(have);
        }
      } else if (seg_have) {
        seg_arrive(la.acc + (uint64_t)seg_owner * 32u, seg_value, 1u, seg_total, p.out + seg_owner, p.done, la.nbuf);
      }
    }
  }
  if (defer && pend_rounds) flush_trailers();
  if (p.verify && blockIdx.x == 0 && tid == 0) atomicAdd(&p.counts[TCRC_CNT_CHECKED], (unsigned long long)n);
  if constexpr (SEG) {
    // What the waves still hold meets in LDS: when it is all for one buffer (the usual case) the workgroup
    // arrives at that buffer's line once.
    __shared__ uint32_t fold[kWavesPerWg][4];
    if (lane == 0) {
      fold[tid >> 6][0] = pend_owner;
      fold[tid >> 6][1] = pend_value;
      fold[tid >> 6][2] = pend_count;
      fold[tid >> 6][3] = pend_total;
    }
    __syncthreads();
    if (tid < 64) {
      const int w = lane < kWavesPerWg ? lane : 0;
      const uint32_t owner = fold[w][0], total = fold[w][3];
      uint32_t v = fold[w][1], c = lane < kWavesPerWg ? fold[w][2] : 0u;
      const unsigned long long act = __ballot(c != 0u);
      if (act) {
        const int first = __ffsll((long long)act) - 1;
        const uint32_t first_owner = __shfl(owner, first), first_total = __shfl(total, first);
        if (__all(c == 0u || owner == first_owner)) {
          if (c == 0u) v = 0u;
#pragma unroll
          for (int o = kWavesPerWg / 2; o >= 1; o >>= 1) {
            v ^= __shfl_xor(v, o);
            c += __shfl_xor(c, o);
          }
          if (lane == 0)
            seg_arrive(la.acc + (uint64_t)first_owner * 32u, v, c, first_total, p.out + first_owner, p.done,
                       la.nbuf);
        } else if (c != 0u) {
          seg_arrive(la.acc + (uint64_t)owner * 32u, v, c, total, p.out + owner, p.done, la.nbuf);
        }
      }
    }
  }
  if constexpr (!SEG) {
    if (p.done.flag) {
      // Every wave's stores have been acknowledged by the L2 at the barrier; a system-scope release by one
      // thread then pushes them out of this XCD's L2 (it may still hold them, and the flag takes another of
      // its channels).  The workgroups count themselves off; the last one tells the host.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        bool last = gridDim.x == 1u;
        if (!last) {
          last = __hip_atomic_fetch_add(p.done.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x;
          if (last) (void)__hip_atomic_exchange(p.done.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(p.done.flag, p.done.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---- log scan --------------------------------------------------------------------------------------------
// One launch verifies a slice of a log image the way log::Reader::ReadPhysicalRecord does block by block
// (db/log_reader.cc:314-406).  The image is dealt to the workgroups in chunks of consecutive 32 KiB blocks.
// Wave 0 of a workgroup walks the record headers - every lane one block, the structural checks of
// db/log_reader.cc:317-384 - and cuts every checksummed span, from its end, into pieces of at most kWalPiece
// bytes; one 8-byte entry per piece goes into a ring in LDS, densely packed, eight entries to a round.  The
// other fifteen waves (and wave 0 itself, whenever the ring is full and once the walk is over) take rounds
// from the ring and checksum them with eight lanes per entry.  A round carries a sequence number: the walker
// may write round t when it reads t, sets it to t + 1 when the round is complete, and whoever took ticket t
// waits for t + 1, copies the entries out and sets t + ring size - so the walk of the next blocks and the
// checksums of the current ones run side by side without a barrier.
// A span of one piece is compared with its stored checksum right away.  The pieces of a longer span (a
// fragment can be 32 KiB: sixteen pieces, checksummed by different waves at the same time) meet in a slot in
// LDS: every piece's remainder is multiplied by x^(8 * bytes behind it), XORed into the slot and counted off;
// whoever counts off last compares the sum with the stored checksum, which the first piece left in the slot,
// and frees the slot.
// Per block the kernel writes (n_records, first_bad, end_offset, end_code); the reference drops the rest of a
// block at its first bad checksum (db/log_reader.cc:391-397), so that is all its policy code needs.
// Every wait in the kernel is bounded: should the protocol ever stall, the workgroup gives up, and the
// launch reports 2^62 mismatches instead of hanging the GPU.
constexpr uint32_t kWalRingRounds = 176;                  // rounds of 8 entries in the ring (11 KiB)
constexpr int kWalSlots = 63;                             // spans of several pieces in flight (id 63 means "none")
constexpr uint32_t kWalStepRounds = 64;                   // at most this many rounds are written per walk step
constexpr uint32_t kWalSpinLimit = 1u << 21;              // bounded waits (about a second)
// entry: x = block (19 bits) | pieces behind this one << 19 (5 bits) | first piece of its span << 24 | valid << 25 |
//            slot << 26 (63: the span is this one piece)
//        y = start inside the block (15 bits) | length << 15
constexpr uint32_t kWalEntryFirst = 1u << 24, kWalEntryValid = 1u << 25, kWalNoSlot = 63u;
struct WalSlot {
  uint32_t acc;        // XOR of the pieces' shifted remainders so far
  uint32_t remaining;  // 0: free; else pieces still to come + 1
  uint32_t stored;     // the span's stored checksum, left here by whoever checksums the first piece
  uint32_t blk;        // where the span is: its block and the offset of its first byte in it
  uint32_t start;
};

// lanes below this one whose bit is set in `mask`
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t lds_load(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void k_wal_fused(WalScanParams p) {
  constexpr int G = 8;
  constexpr int kAuxDwords = AuxDwords<G>::value;
  __shared__ __align__(16) uint32_t lds[kMainDwords + kAuxDwords];
  __shared__ uint2 ring[kWalRingRounds * 8];
  __shared__ uint32_t seq[kWalRingRounds];
  __shared__ WalSlot slots[kWalSlots];
  // [0] next ticket, [1] rounds in all once the walk is over (till then ~0), [2] set when a wait ran out
  __shared__ uint32_t ctl[4];
  __shared__ uint32_t pow_piece[64];  // [q * 4 + l]: x^(8 * kWalPiece * q) * x^(8 l)

  const int tid = threadIdx.x;
  // trace (development aid): wave 0 notes the time at [0] start, [1] tables in place, [2] walk over, [31] end
  auto stamp = [&](uint32_t at) {
    if (p.trace && tid == 0 && at < 32u) p.trace[blockIdx.x * 32u + at] = wall_clock64();
  };
  stamp(0);
  fill_main_tables(lds, p.tab_main, tid);
  static_assert(kAuxDwords % 4 == 0, "the small sets are copied sixteen bytes at a time");
  for (int q = tid; q < kAuxDwords / 4; q += kThreads)
    reinterpret_cast<uint4*>(lds + kMainDwords)[q] = reinterpret_cast<const uint4*>(p.tab_aux)[q];
  for (uint32_t i = tid; i < kWalRingRounds; i += kThreads) seq[i] = i;
  for (int i = tid; i < kWalSlots; i += kThreads) slots[i].remaining = 0u;
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 64; ++i) pow_piece[i] = p.pow_piece[i];
    ctl[0] = 0u;
    ctl[1] = 0xffffffffu;
    ctl[2] = 0u;
  }
  __syncthreads();
  stamp(1);

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & (G - 1);
  const int grp = lane / G;
  const uint32_t LB = (uint32_t)(lane & 31) * 4u | 0x10000u;

  // This workgroup's blocks: the image is dealt out in chunks of 2^chunk_shift consecutive blocks, chunk c to
  // workgroup c mod gridDim.x, so that at any time the workgroups read neighbouring pieces of the image.  The
  // walker numbers its own blocks 0, 1, 2, ...: number v is block v mod 2^s of the workgroup's chunk v >> s.
  // (Handing chunks out from a counter in device memory instead was tried twice.  Taken as fast as the walker
  // can: nothing changes for a large image, whose scan is bound by the whole chip's memory system and not by
  // the slowest workgroup, and for a small one the workgroups that start first take everything.  Taken only
  // when a workgroup runs short of rounds: the walk of a block is a chain of dependent loads of more than a
  // microsecond each, and it is the walker's long lead over the checksums that hides it - on demand the chains
  // land on the critical path: 402 instead of 348 us for the 2 GB log of 2 KiB records, 91 instead of 73 us for
  // 256 MiB of mixed sizes.)
  const uint32_t cs = p.chunk_shift;
  const uint32_t nchunks = (p.nblocks + (1u << cs) - 1u) >> cs;
  const uint32_t my_chunks = nchunks > blockIdx.x ? (nchunks - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
  const uint32_t b_hi = my_chunks << cs;  // numbers, not all of them blocks of the image (the last chunk may be short)
  auto block_of = [&](uint32_t v) -> uint32_t {
    return (((v >> cs) * gridDim.x + blockIdx.x) << cs) + (v & ((1u << cs) - 1u));
  };

  // walker state (wave 0): every lane walks one block at a time
  uint32_t w_next = 0;   // next number to hand to an idle lane (the same in every lane)
  uint32_t w_blk = 0, w_pos = 0, w_k = 0, w_records = 0;
  bool w_active = false;
  uint32_t w_count = 0;  // entries written to the ring so far (the same in every lane)
  uint32_t w_open = 0;   // rounds [0, w_open) are known to be the walker's to write or already written
  uint32_t w_spans = 0;  // spans of several pieces so far: span n uses slot n mod kWalSlots
  bool w_done = false;   // the walk is over and the last round is out

  // One step of the walk: every active lane looks at the record under its cursor and, if there is room, puts
  // the record's pieces into the ring.  Returns true if the ring (or the slots) had no room for some record.
  // `hdr` is the header under the lane's cursor, fetched by the step before (or by the caller, for a lane that
  // has none yet); the step leaves the next one in it.
  auto block_end = [&](uint32_t blk) -> uint64_t {
    const uint64_t e = (uint64_t)blk * kLogBlockBytes + kLogBlockBytes;
    return e < p.size ? e : p.size;
  };
  auto walk_step = [&](WalHeader& hdr) -> bool {
    const unsigned long long idle = __ballot(!w_active);
    if (idle && w_next < b_hi) {
      const uint32_t nv = w_next + lanes_below(idle);
      const uint32_t nb = block_of(nv);
      if (!w_active && nv < b_hi && nb < p.nblocks) {
        w_active = true;
        w_blk = nb;
        w_pos = 0;
        w_k = 0;
        p.blocks[nb].first_bad = 0xffffffffu;  // lowered by whoever finds a bad checksum in the block
        hdr = wal_fetch(p.img, (uint64_t)nb * kLogBlockBytes, block_end(nb));
      }
      const uint32_t handed = w_next + (uint32_t)__popcll(idle);
      w_next = handed < b_hi ? handed : b_hi;
    }
    if (!__any(w_active)) {
      // Nothing left to walk: the last round is padded with empty entries and goes out, and the number of
      // rounds is made known.  (The round is the walker's: w_open covers it, see below.)
      const uint32_t pad = (8u - (w_count & 7u)) & 7u;
      if (pad) {
        if ((uint32_t)lane < pad) ring[(w_count + lane) % (kWalRingRounds * 8u)] = make_uint2(0u, 0u);
        if (lane == 0) __hip_atomic_store(&seq[(w_count >> 3) % kWalRingRounds], (w_count >> 3) + 1u, __ATOMIC_RELEASE,
                                          __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (lane == 0) lds_store(&ctl[1], (w_count + 7u) >> 3);
      w_done = true;
      return false;
    }
    const uint64_t base = (uint64_t)w_blk * kLogBlockBytes;
    const uint64_t end = base + kLogBlockBytes < p.size ? base + kLogBlockBytes : p.size;
    const bool last = p.tail && end == p.size;
    uint32_t hs = 0, len = 0, code = 0;
    bool ok = false;
    if (w_active) ok = wal_decode(hdr, base + w_pos, end, last, p.log_number, &hs, &len, &code);
    // The next header is asked for now, before the room for this record is worked out: the walk of a block is a
    // chain of dependent loads, and the bookkeeping below takes about as long as one of them.  (Should the
    // record find no room, the lane keeps its header and the fetch was for nothing.)
    WalHeader nxt = {0u, 0u};
    if (ok) nxt = wal_fetch(p.img, base + w_pos + hs + len, end);
    if (w_active && !ok) {  // the block's walk ends here: that needs no room in the ring
      tcrc_wal_block* b = p.blocks + w_blk;
      b->n_records = w_k;
      b->end_offset = w_pos;
      b->end_code = code;
      w_records += w_k;
      w_active = false;
    }
    // the CRC covers the type byte, the log number if any, and the payload
    const uint32_t span = len + hs - 6u;
    const uint32_t c = ok ? (span + kWalPiece - 1u) / kWalPiece : 0u;  // 1..16 pieces
    const bool multi = c > 1u;
    // Room: rounds from the one w_count points into onwards are the walker's once their sequence number says
    // so.  Only when what is known to be free does not take this step's entries are the sequence numbers looked
    // at again: lane i looks at round w_open + i; the rounds up to the first one that is still somebody's are
    // added.
    // Counts and places come from ballots - one per bit of the piece count, and a count of the lower lanes in it:
    // scalar and cheap vector instructions only, where a scan with shuffles takes six trips through the LDS
    // crossbar (the walker's steps are a chain of dependent operations, so their latency is what counts).  A step
    // in which no record is longer than a piece - every step of a log of small records - gets by with one ballot
    // and does not look at the slots.
    const unsigned long long m_multi = __ballot(multi);
    uint32_t step_entries = 0, below_e = 0;
    if (m_multi == 0ull) {
      const unsigned long long m_ok = __ballot(ok);
      step_entries = (uint32_t)__popcll(m_ok);
      below_e = lanes_below(m_ok);
    } else {
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const unsigned long long mb = __ballot((c >> b) & 1u);
        step_entries += (uint32_t)__popcll(mb) << b;
        below_e += lanes_below(mb) << b;
      }
    }
    if (step_entries > w_open * 8u - w_count) {
      const uint32_t t = w_open + (uint32_t)lane;
      const bool mine = (uint32_t)lane < kWalStepRounds && lds_load(&seq[t % kWalRingRounds]) == t;
      const unsigned long long m = __ballot(mine);
      w_open += m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
      if (w_open > (w_count >> 3) + kWalStepRounds) w_open = (w_count >> 3) + kWalStepRounds;
    }
    const uint32_t room = w_open * 8u - w_count;
    // Lanes take room in lane order; those that no longer fit keep their record for the next step.  A span of
    // several pieces also needs its slot to be free.
    const uint32_t upto_e = below_e + c;
    const uint32_t my_span = w_spans + lanes_below(m_multi);
    bool before_busy = true;
    if (m_multi != 0ull) {
      const bool slot_busy = multi && lds_load(&slots[my_span % kWalSlots].remaining) != 0u;
      const unsigned long long m_busy = __ballot(slot_busy || (multi && my_span - w_spans >= (uint32_t)kWalSlots));
      before_busy = m_busy == 0ull || lane < __builtin_ctzll(m_busy);
    }
    const bool fits = ok && upto_e <= room && before_busy;
    if (fits) {
      const uint32_t slot = multi ? my_span % kWalSlots : kWalNoSlot;
      const uint32_t start = w_pos + 6u, stop = start + span;
      const uint32_t at = w_count + upto_e - c;
      for (uint32_t i = 0; i < c; ++i) {
        const uint32_t q = c - 1u - i;  // pieces behind piece i
        const uint32_t pe = stop - q * kWalPiece;
        const uint32_t ps = i == 0u ? start : pe - kWalPiece;
        ring[(at + i) % (kWalRingRounds * 8u)] =
            make_uint2(w_blk | (q << 19) | (i == 0u ? kWalEntryFirst : 0u) | kWalEntryValid | (slot << 26), ps | ((pe - ps) << 15));
      }
      if (multi) {
        slots[slot].acc = 0u;
        lds_store(&slots[slot].remaining, c + 1u);
      }
      w_pos += hs + len;
      ++w_k;
      hdr = nxt;
    }
    const unsigned long long m_fit = __ballot(fits);
    if (m_fit) {
      const int top = 63 - __builtin_clzll(m_fit);
      const uint32_t new_count = w_count + (uint32_t)__shfl((int)upto_e, top);
      w_spans += (uint32_t)__popcll(__ballot(fits && multi));
      // rounds that are complete now go out: lane i publishes round (old count / 8) + i
      const uint32_t r0 = w_count >> 3, r1 = new_count >> 3;
      if ((uint32_t)lane < r1 - r0)
        __hip_atomic_store(&seq[(r0 + lane) % kWalRingRounds], r0 + lane + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      w_count = new_count;
    }
    return __any(ok && !fits);
  };

  // A checksum that does not match: counted, and the record's index inside its block - found by walking the
  // block's headers again, this is rare - lowers the block's first_bad.
  auto mismatch = [&](uint32_t blk, uint32_t start) {
    atomicAdd(&p.counts[TCRC_CNT_MISMATCH], 1ull);
    atomicMin(&p.counts[TCRC_CNT_FIRST_BAD], (unsigned long long)(p.block_base + blk));
    const uint64_t base = (uint64_t)blk * kLogBlockBytes;
    const uint64_t end = base + kLogBlockBytes < p.size ? base + kLogBlockBytes : p.size;
    const bool last = p.tail && end == p.size;
    uint32_t pos = 0, k = 0, hs = 0, ln = 0, code = 0;
    while (pos + 6u < start && wal_step(p.img, base + pos, end, last, p.log_number, &hs, &ln, &code)) {
      pos += hs + ln;
      ++k;
    }
    atomicMin(&p.blocks[blk].first_bad, k);
  };

  uint32_t spins = 0;
  for (;;) {
    if (lds_load(&ctl[2])) break;  // some wave of the workgroup gave up waiting
    uint32_t t = 0;
    if (wave == 0 && !w_done) {
      __builtin_amdgcn_s_setprio(3);  // the walk is the critical path of a workgroup with little to do
      // The walker.  While its records find room it only walks; when they do not, it takes a round itself if one
      // is out that nobody has taken (so that it never waits for a round only it could write).
      // (the headers under the cursors are fetched again whenever the walker comes back from other work)
      WalHeader hdr = {0u, 0u};
      if (w_active) hdr = wal_fetch(p.img, (uint64_t)w_blk * kLogBlockBytes + w_pos, block_end(w_blk));
      bool blocked = false;
      do {
        blocked = walk_step(hdr);
      } while (!blocked && !w_done && !lds_load(&ctl[2]));
      if (!blocked) {
        if (w_done) {
          stamp(2);
          __builtin_amdgcn_s_setprio(0);
        }
        continue;
      }
      __builtin_amdgcn_s_setprio(0);  // about to checksum a round (or to wait): like everybody else
      uint32_t got = 0xffffffffu;
      if (lane == 0) {
        const uint32_t out = w_count >> 3;  // rounds that are out
        uint32_t h = lds_load(&ctl[0]);
        while (h < out && !__hip_atomic_compare_exchange_strong(&ctl[0], &h, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) {
        }
        if (h < out) got = h;
      }
      got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
      if (got == 0xffffffffu) {  // all rounds that are out are being worked on: their slots come back soon
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kWalSpinLimit) lds_store(&ctl[2], 1u);
        continue;
      }
      t = got;
    } else {
      if (lane == 0) t = atomicAdd(&ctl[0], 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      bool over = false;
      for (;;) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)__hip_atomic_load(&seq[t % kWalRingRounds], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (s == t + 1u) break;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_load(&ctl[1]));
        if (t >= total || lds_load(&ctl[2])) {
          over = true;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kWalSpinLimit) {
          lds_store(&ctl[2], 1u);
          over = true;
          break;
        }
      }
      if (over) break;
    }
    spins = 0;
    // round t: the entries are copied out, and the round is handed back to the walker
    const uint2 e = ring[(t % kWalRingRounds) * 8u + grp];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane has its entry before the round is given away
    if (lane == 0)
      __hip_atomic_store(&seq[t % kWalRingRounds], t + kWalRingRounds, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool valid = (e.x & kWalEntryValid) != 0u;
    const bool first = (e.x & kWalEntryFirst) != 0u;
    const uint32_t blk = e.x & 0x7ffffu, q = (e.x >> 19) & 31u, slot = e.x >> 26;
    const uint32_t start = e.y & 0x7fffu, len = e.y >> 15;
    const uint8_t* ptr = p.img + (uint64_t)blk * kLogBlockBytes + start;
    // the stored value is fetched now so its latency hides behind the row loop
    uint32_t stored = 0;
    if (valid && first && j == G - 1) __builtin_memcpy(&stored, ptr - 6, 4);
    const uint32_t core = span_remainder<G, NT>(lds, LB, lane, ptr, len, first ? 0u : 0xffffffffu, valid,
                                                (const uint8_t*)p.tab_main);
    // A piece with q pieces behind it enters its span's remainder multiplied by x^(8 * kWalPiece * q).  The 32
    // shift-and-xor steps of that multiplication are split over four lanes of the group, like the un-shift in
    // span_remainder: lane l takes bits [8 l, 8 l + 8) of the remainder against the multiplier times x^(8 l),
    // which comes from the table.  (One lane doing all 32 steps costs the whole wave four times as long.)
    uint32_t shifted = core;
    if (__any(valid && q != 0u)) {
      const uint32_t rr = __shfl(core, lane | (G - 1));
      const int l = j & 3;
      uint32_t b = pow_piece[(q & 15u) * 4u + (uint32_t)l];
      uint32_t part = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        part ^= ((rr >> (31 - 8 * l - k)) & 1u) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? 0x82f63b78u : 0u);
      }
      part ^= __shfl_xor(part, 1);
      part ^= __shfl_xor(part, 2);
      if (q != 0u) shifted = part;
    }
    if (valid && j == G - 1) {
      if (slot == kWalNoSlot) {
        const uint32_t crc = ~core;
        if (stored != ((crc >> 15) | (crc << 17)) + kMaskDelta) mismatch(blk, start);
      } else {
        WalSlot* sl = &slots[slot];
        if (first) {
          lds_store(&sl->stored, stored);
          lds_store(&sl->blk, blk);
          lds_store(&sl->start, start);
        }
        const uint32_t v = shifted;
        (void)__hip_atomic_fetch_xor(&sl->acc, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // release what this piece left in the slot, acquire what the others left
        const uint32_t before = __hip_atomic_fetch_sub(&sl->remaining, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (before == 2u) {  // the last piece: compare, then give the slot back
          const uint32_t crc = ~lds_load(&sl->acc);
          const uint32_t want = lds_load(&sl->stored);
          const uint32_t sblk = lds_load(&sl->blk), sstart = lds_load(&sl->start);
          __hip_atomic_store(&sl->remaining, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (want != ((crc >> 15) | (crc << 17)) + kMaskDelta) mismatch(sblk, sstart);
        }
      }
    }
  }
  stamp(31);
  if (wave == 0) {
    uint32_t total = w_records;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0) {
      if (total) atomicAdd(&p.counts[TCRC_CNT_CHECKED], (unsigned long long)total);
      if (lds_load(&ctl[2])) atomicAdd(&p.counts[TCRC_CNT_MISMATCH], 1ull << 62);  // a wait ran out: say so loudly
      if (p.records_host) {
        // The launch's record count goes to the host without a copy command: workgroups add theirs to a device
        // word and count themselves off; the last one takes the sum out (leaving both words zero for the next
        // launch) and adds it to the caller's word in pinned host memory.  The additions are acknowledged
        // before the count-off is issued (see seg_arrive).
        unsigned long long* w = p.launch_words;
        (void)__hip_atomic_fetch_add(w, (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long before = __hip_atomic_fetch_add(w + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1ull == (unsigned long long)gridDim.x) {
          const unsigned long long sum = __hip_atomic_exchange(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          (void)__hip_atomic_exchange(w + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *p.records_host += sum;  // launches of one stream run one after the other
        }
      }
    }
  }
}


// ---- log scan, walker and placer apart ------------------------------------------------------------------
// The same scan with the header walk cut down to what is sequential.  In k_wal_fused one dependent step per record
// decodes the header, works out the record's pieces, finds room in the ring and slots for them and writes them;
// here wave 0 only chases headers (fetch, the structural checks of db/log_reader.cc:317-384, advance) and every
// lane appends one 8-byte (block, start, length) entry per checksummed span to a short queue of its own in LDS -
// no ballot, no shared counter, and a lane whose queue is full waits by itself.  Wave 1 takes up to 64 entries of
// those queues per step, dealt over the queues that have any - parallel over records, no header chain in its way -
// and does what is left:
// piece counts, places in the ring from ballots, slots for spans of several pieces, the ring writes.  Both go on
// to checksum rounds like the other fourteen waves when they have nothing else to do.
constexpr uint32_t kSplitRingRounds = 32;    // rounds of 8 entries in the ring (2 KiB)
constexpr uint32_t kSplitListEntries = 1024;  // spans the walker may be ahead of the placer (8 KiB), shared out among its lanes' queues
constexpr uint32_t kSplitHelpMin = 12;       // walker and placer take a round themselves only with this many waiting
// Small records: how far in front of the walker's cursor the placer asks for the image's 64-byte sectors, up to
// which span length a record counts as small, and the number of waiting rounds from which on it stops asking
constexpr uint32_t kSplitAhead = 1024, kSplitAheadSpan = 640, kSplitAheadBacklog = 16;
// classes of pieces that share rounds: longer than kSplitLong bytes, longer than kSplitShort, and the rest
constexpr uint32_t kSplitLong = 1024, kSplitShort = 256;

// TRACE: the instantiation that writes phase timestamps (development aid; the time stamps' registers are enough to
// change how the compiler schedules the checksum loops, so the ordinary instantiation has none)
template <bool NT, bool TRACE>
__global__ __launch_bounds__(kThreads) void k_wal_split(WalScanParams p) {
  constexpr int G = 8;
  constexpr int kAuxDwords = AuxDwords<G>::value;
  constexpr uint32_t RR = kSplitRingRounds;
  __shared__ __align__(16) uint32_t lds[kMainDwords + kAuxDwords];
  __shared__ uint2 ring[RR * 8];
  __shared__ uint32_t seq[RR];
  __shared__ uint2 rlist[kSplitListEntries];
  __shared__ WalSlot slots[kWalSlots];
  // [0] next ticket, [1] rounds in all once everything is placed (till then ~0), [2] set when a wait ran out,
  // [3] rounds that are out, [4] bit 31: the walk is over, [6] candidate records of the blocks walked so far
  __shared__ uint32_t ctl[8];
  // The walker's lanes' queues, in rlist: the workgroup's first Q = 2^qs lanes walk (Q: the number of its blocks
  // rounded up to a power of two, at most 64), queue i has 1024 / Q places, and entry number e of queue i lies at
  // ((e mod places) << qs) + i.  An entry: .x = block, .y = start | length << 15.
  // qctr[i]: .x entries the walker's lane i has put into its queue (written behind the entry: the LDS performs a
  // wave's operations in order), .y entries the placer has taken from it; both mod 65536.
  // qmap: the queues that have entries, listed by the placer for its step.
  __shared__ uint2 qctr[64];
  __shared__ uint8_t qmap[64];
  __shared__ uint32_t pow_piece[64];  // [q * 4 + l]: x^(8 * kWalPiece * q) * x^(8 l)
  __shared__ uint32_t ahead_sink[64];  // where the sectors asked for ahead of the walker land (never read)
  // The walker's cursors and counters (wave 0) and the placer's counters (wave 1) while those waves are away on a
  // round: kept here and not in registers, so that the checksum code of all sixteen waves has the registers to
  // itself (with them live across it, the compiler schedules the row loop's table lookups four at a time instead
  // of sixteen, and a scan that waits for its checksums is 3 % slower).
  __shared__ uint4 w_save[64];
  __shared__ uint32_t u_save[8];

  const int tid = threadIdx.x;
  // trace (development aid): [0] start, [1] tables in place, [2] walk over, [3] everything placed, [31] end of wave 0
  auto stamp = [&](uint32_t at, int who) {
    if constexpr (TRACE) {
      if (p.trace && tid == who && at < 32u) p.trace[blockIdx.x * 32u + at] = wall_clock64();
    } else {
      (void)at;
      (void)who;
    }
  };
  stamp(0, 0);
  fill_main_tables(lds, p.tab_main, tid);
  for (int q = tid; q < kAuxDwords / 4; q += kThreads)
    reinterpret_cast<uint4*>(lds + kMainDwords)[q] = reinterpret_cast<const uint4*>(p.tab_aux)[q];
  for (uint32_t i = tid; i < RR; i += kThreads) seq[i] = i;
  for (int i = tid; i < kWalSlots; i += kThreads) slots[i].remaining = 0u;
  if (tid < 64) pow_piece[tid] = p.pow_piece[tid];
  if (tid < 8) ctl[tid] = tid == 1 ? 0xffffffffu : 0u;
  if (tid < 64) w_save[tid] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < 8) u_save[tid] = 0u;
  if (tid < 64) qctr[tid] = make_uint2(0u, 0u);
  __syncthreads();
  stamp(1, 0);

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (said to be the same in every lane: the branches on it are scalar then)
  const int j = lane & (G - 1);
  const int grp = lane / G;
  const uint32_t LB = (uint32_t)(lane & 31) * 4u | 0x10000u;

  // this workgroup's blocks: as in k_wal_fused
  const uint32_t cs = p.chunk_shift;
  const uint32_t nchunks = (p.nblocks + (1u << cs) - 1u) >> cs;
  const uint32_t my_chunks = nchunks > blockIdx.x ? (nchunks - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
  const uint32_t b_hi = my_chunks << cs;
  auto block_of = [&](uint32_t v) -> uint32_t {
    return (((v >> cs) * gridDim.x + blockIdx.x) << cs) + (v & ((1u << cs) - 1u));
  };
  auto block_end = [&](uint32_t blk) -> uint64_t {
    const uint64_t e = (uint64_t)blk * kLogBlockBytes + kLogBlockBytes;
    return e < p.size ? e : p.size;
  };
  auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  // the lanes in which a condition holds, straight from the comparison (__ballot and __any take an int: the
  // condition is first made a number and then compared with zero again, and the walker's step counts instructions)
  auto lanes_where = [](bool c) -> unsigned long long { return __builtin_amdgcn_ballot_w64(c); };
  // the queues' geometry
  const uint32_t lanes_used = b_hi < 64u ? b_hi : 64u;
  const uint32_t qs = lanes_used <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(lanes_used - 1u);  // log2 of the queues
  const uint32_t cap_shift = 10u - qs;                                                         // log2 of a queue's places
  const uint32_t cap = 1u << cap_shift;
  auto lds_acquire = [](const uint32_t* q) -> uint32_t {
    return __hip_atomic_load(q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto lds_release = [](uint32_t* q, uint32_t v) {
    __hip_atomic_store(q, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  // ---- the walker (wave 0): every lane chases the headers of one block at a time
  uint32_t w_next = 0;  // next number to hand to an idle lane (the same in every lane)
  uint32_t w_blk = 0, w_pos = 0, w_k = 0;
  uint32_t w_len = 0;          // bytes of the lane's block that are in the image (32768 but for the image's last block)
  auto w_ptr = [&]() -> const uint8_t* { return p.img + (uint64_t)w_blk * kLogBlockBytes; };  // the block's first byte
  // 0: the lane has no block; 1: it has one, and the header under its cursor has to be fetched and decoded the
  // general way (fewer than twelve bytes are left in the block); 3: it has one, and w_hdr holds header bytes 4..11
  // under the cursor, fetched with one 8-byte load
  uint32_t w_state = 0;
  bool w_done = false;
  uint32_t w_seq = 0;  // entries the lane has put into its queue so far (mod 65536 where it is kept)
  const uint32_t ln32 = (uint32_t)p.log_number;
  const bool ln_fits = (p.log_number >> 32) == 0;  // a recyclable header holds 32 bits: a larger log number never matches
  // One step: every lane with a block looks at the header under its cursor (fetched by the step before) and puts
  // the record's checksummed span into its queue.  Returns true if some lane's queue had no room.
  // The step is one wave's chain of dependent instructions and nothing else bounds a log of small records, so it
  // is written for few instructions and few branches: the common case - at least twelve bytes left in the block, a
  // header that passes every check of db/log_reader.cc:317-384 - is decoded in 32-bit arithmetic relative to the
  // block with the conditions combined bitwise (with && the compiler branches around every operand); whatever else
  // a lane meets (the last bytes of a block, a header that ends the block's walk) takes the general wal_fetch /
  // wal_decode, which the whole wave then steps through once.
  // walk_run: steps until the walk is over (returns 2), some lane's queue is full (1) or, looked at every 32 steps,
  // some wave of the workgroup has given up waiting (0).  hdr: header bytes 4..11 under every cursor in state 3.
  auto walk_run = [&](WalHeader hdr) -> uint32_t {
   uint32_t steps = 0;
   for (;;) {
    const uint32_t tk = lds_load(&qctr[lane].y);  // asked for now, looked at when the record is known to be good
    if (w_next < b_hi) {
      const unsigned long long idle = lanes_where(w_state == 0u);
      if (idle) {
        const uint32_t nv = w_next + lanes_below(idle);
        const uint32_t nb = block_of(nv);
        if (w_state == 0u && nv < b_hi && nb < p.nblocks) {
          w_blk = nb;
          w_pos = 0;
          w_k = 0;
          const uint64_t bbase = (uint64_t)nb * kLogBlockBytes;
          w_len = (uint32_t)(block_end(nb) - bbase);
          p.blocks[nb].first_bad = 0xffffffffu;  // lowered by whoever finds a bad checksum in the block
          w_state = w_len >= 12u ? 3u : 1u;
          if (w_state == 3u) {
            uint2 w;
            __builtin_memcpy(&w, w_ptr() + 4, 8);
            hdr.lo = w.x;
            hdr.hi = w.y;
          }
        }
        const uint32_t handed = w_next + (uint32_t)__popcll(idle);
        w_next = handed < b_hi ? handed : b_hi;
      }
    }
    uint32_t len = hdr.lo & 0xffffu;
    const uint32_t type = (hdr.lo >> 16) & 0xffu;
    const bool recyc = type - 5u < 4u;
    uint32_t hs = recyc ? kLogRecyclableHeader : kLogHeader;
    const uint32_t lognum = __builtin_amdgcn_alignbyte(hdr.hi, hdr.lo, 3);  // header bytes 7..10
    uint32_t npos = w_pos + hs + len;
    // The chain of a block's walk is: header -> where the next one lies -> its load.  So the load goes out here,
    // before anything is checked - from inside the block whatever the header holds (a position past the place of
    // the block's last whole header is pulled back to it; what comes back is then not used).  A lane whose queue
    // is full asks for the header under its cursor once more instead: it stays where it is for a step while the
    // others go on.  The checks, the queue and the bookkeeping below run while the loads are under way.
    const bool room = ((w_seq - tk) & 0xffffu) < cap;
    uint32_t at = room ? npos : w_pos;
    at = at < w_len - 12u ? at : w_len - 12u;  // (lanes in state 3 have twelve bytes and more)
    WalHeader nxt;
    {
      const uint8_t* from = w_state == 3u ? w_ptr() + at + 4u : (const uint8_t*)p.tab_main;  // others read a harmless place
      uint2 w;
      __builtin_memcpy(&w, from, 8);
      nxt.lo = w.x;
      nxt.hi = w.y;
    }
    bool ok = (w_state == 3u) & (npos <= w_len) & ((hdr.lo & 0xffffffu) != 0u) & (!recyc | ((lognum == ln32) & ln_fits));
    bool go = ok & room;
    // One question settles the usual step: has every lane that has a block a good record and room for it?  Only
    // if not does the wave go through what else can be (a block's end, a bad header, a full queue, no blocks left).
    const bool active = w_state != 0u;
    if (lanes_where(active & !go) != 0ull || lanes_where(active) == 0ull) {
      const bool odd = active & !ok;
      if (lanes_where(odd) != 0ull) {
        if (odd) {
          const uint64_t base = (uint64_t)w_blk * kLogBlockBytes;
          const uint64_t end = base + w_len;
          const bool last = p.tail && end == p.size;
          uint32_t code = 0;
          WalHeader h = hdr;
          if (w_state != 3u) h = wal_fetch(p.img, base + w_pos, end);
          ok = wal_decode(h, base + w_pos, end, last, p.log_number, &hs, &len, &code);
          npos = w_pos + hs + len;
          if (!ok) {  // the block's walk ends here
            tcrc_wal_block* b = p.blocks + w_blk;
            b->n_records = w_k;
            b->end_offset = w_pos;
            b->end_code = code;
            atomicAdd(&ctl[6], w_k);  // the workgroup's candidate records
            w_state = 0u;
          }
        }
      }
      // from here on a lane has a block exactly if its record is good
      if (lanes_where(ok) == 0ull && w_next >= b_hi) {
        if (lane == 0) lds_store(&ctl[4], 0x80000000u);  // behind the last entry: the LDS performs a wave's operations in order
        w_done = true;
        return 2u;
      }
      go = ok & room;
      if (lanes_where(ok & !room) != 0ull) {
        // Some queue is full.  If rounds are waiting for checksum waves, the scan is not waiting for the walk: the
        // walker goes and takes one.  If not - a log of small records, the other waves idle - it walks on with
        // the lanes that can, unless none can.  (What the lanes have asked for is dropped: the walker fetches the
        // headers under its cursors when it comes back.)
        if (lanes_where(go) == 0ull) return 1u;
        const uint32_t out = uni(lds_load(&ctl[3])), tickets = uni(lds_load(&ctl[0]));
        if ((int32_t)(out - tickets) >= (int32_t)kSplitHelpMin) return 1u;
      }
    }
    const bool fastn = ok & (w_len - (go ? npos : w_pos) >= 12u);  // what was asked for above is the header the next step needs
    if (go) {
      // the CRC covers the type byte, the log number if any, and the payload: [pos + 6, pos + hs + len)
      rlist[((w_seq & (cap - 1u)) << qs) + (uint32_t)lane] = make_uint2(w_blk, (w_pos + 6u) | ((len + hs - 6u) << 15));
      ++w_seq;
      asm volatile("" ::: "memory");  // the count goes out behind the entry
      lds_store(&qctr[lane].x, w_seq);
      w_pos = npos;
      ++w_k;
    }
    if (ok) w_state = fastn ? 3u : 1u;
    hdr = nxt;
    if ((++steps & 31u) == 0u && uni(lds_load(&ctl[2])) != 0u) return 0u;
   }
  };

  // ---- the placer (wave 1): up to 64 listed spans per step
  uint32_t p_count = 0;  // entries written to the ring so far
  uint32_t p_open = 0;   // rounds [0, p_open) are known to be the placer's to write or already written
  uint32_t p_spans = 0;  // spans of several pieces so far: span n uses slot n mod kWalSlots
  bool p_done = false;
  // Returns 0 when every span it took found room, 1 when the ring or the slots had no room for some, 2 when the
  // queues were empty.
  // Experiment: rounds the placer may be ahead of the tickets handed out (0: as many as the ring holds)
  const uint32_t lead_max = (p.tune >> 16) & 0xffu;
  auto lead_cap = [&](uint32_t open) -> uint32_t {
    if (lead_max == 0u) return open;
    const uint32_t lim = uni(lds_load(&ctl[0])) + lead_max;
    return open < lim ? open : lim;
  };
  auto place_step = [&]() -> int {
    // One look at the ring before anything else: with less than a round's worth of room the step could place
    // next to nothing, and all it would do is work out that it is blocked (a scan that waits for its checksum
    // waves comes here after every round the placer takes itself).
    if (p_open * 8u - p_count < 8u) {
      const uint32_t t = p_open + (uint32_t)lane;
      const bool mine = (uint32_t)lane < kWalStepRounds && lds_load(&seq[t % RR]) == t;
      const unsigned long long m = __ballot(mine);
      p_open += m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
      if (p_open > (p_count >> 3) + kWalStepRounds) p_open = (p_count >> 3) + kWalStepRounds;
      p_open = lead_cap(p_open);
      if (p_open * 8u - p_count < 8u && (p_count & 7u) == 0u) return 1;
    }
    const uint32_t over = uni(lds_acquire(&ctl[4])) >> 31;  // looked at before the queues: if set, every entry is in
    // The queues that have entries are listed, and the step's 64 lanes dealt over them: lane l takes queue
    // number l mod A of the list, at depth l / A behind the entries taken so far (A queues in the list) - one
    // busy block's queue alone is emptied 64 entries at a step, 64 busy ones one entry each.
    uint32_t avail_q = 0;
    if ((uint32_t)lane < (1u << qs)) {
      const uint2 c = qctr[lane];
      avail_q = (c.x - c.y) & 0xffffu;
    }
    const unsigned long long m_q = lanes_where(avail_q != 0u);
    bool has = false;
    uint32_t qi = 0, eseq = 0;
    uint2 e = make_uint2(0u, 0u);
    if (m_q != 0ull) {
      const uint32_t A = (uint32_t)__popcll(m_q);
      if (avail_q != 0u) qmap[lanes_below(m_q)] = (uint8_t)lane;
      // lane / A without an integer division: with the half added the quotient is at least 1/128 away from every
      // integer, far more than the division's rounding
      const uint32_t qd = (uint32_t)(((float)lane + 0.5f) / (float)A);
      qi = qmap[(uint32_t)lane - qd * A];  // (read behind the writes: one wave's LDS operations are performed in order)
      const uint2 c = qctr[qi];
      has = qd < ((c.x - c.y) & 0xffffu);
      eseq = c.y + qd;
      if (has) e = rlist[((eseq & (cap - 1u)) << qs) + qi];
    }
    const unsigned long long m_has = lanes_where(has);
    if (m_has == 0ull) {
      if (!over) return 2;
      // Everything is placed: the last round is padded with empty entries and goes out, and the number of rounds
      // is made known.  (The round is the placer's: p_open covers a round it has written entries into.)
      const uint32_t pad = (8u - (p_count & 7u)) & 7u;
      if (pad) {
        if ((uint32_t)lane < pad) ring[(p_count + lane) % (RR * 8u)] = make_uint2(0u, 0u);
        if (lane == 0) lds_release(&seq[(p_count >> 3) % RR], (p_count >> 3) + 1u);
      }
      if (lane == 0) {
        lds_store(&ctl[3], (p_count + 7u) >> 3);
        lds_store(&ctl[1], (p_count + 7u) >> 3);
      }
      p_done = true;
      return 0;
    }
    const uint32_t blk = e.x, start = e.y & 0x7fffu, span = e.y >> 15;
    const uint32_t c = has ? (span + kWalPiece - 1u) / kWalPiece : 0u;  // 1..16 pieces
    const bool multi = c > 1u;
    // Small records.  The walk of a block is a chain of dependent loads, and with records of a hundred bytes
    // every one of them is the first touch of its line: a trip to memory per record.  Loads return to a wave in
    // the order it issued them, so the walker cannot ask ahead for itself without waiting for what it asked; the
    // placer can - it sees where every cursor is a moment after the walker has moved it, and asks for the
    // 64-byte sectors that the record just listed brought within kSplitAhead bytes of the cursor (straight into
    // a few words of LDS that nobody reads: the point is the copy the L2 keeps).  Not while rounds pile up:
    // then the checksum waves are what the scan waits for, and they would find the lines evicted again.
    const uint32_t ahead_span = (p.tune >> 24) & 0x7fu ? ((p.tune >> 24) & 0x7fu) * 64u : kSplitAheadSpan;  // (experiment knob)
    if (!(p.tune & 1u) && __any(has && span <= ahead_span)) {
      const uint32_t out = p_count >> 3, taken = uni(lds_load(&ctl[0]));
      if (taken + kSplitAheadBacklog > out) {
        const uint64_t bbase = (uint64_t)blk * kLogBlockBytes;
        const uint32_t blen = (uint32_t)(block_end(blk) - bbase);
        const uint32_t lim = blen >= 4u ? ((blen - 4u) >> 6) + 1u : 0u;  // sectors of the block that hold four readable bytes
        // sectors reached so far: up to (header position + kSplitAhead) / 64; a block's first record starts from nothing
        const uint32_t ahead = (p.tune >> 8) & 0xffu ? ((p.tune >> 8) & 0xffu) * 256u : kSplitAhead;
        const uint32_t from = start == 6u ? 0u : ((start - 6u + ahead) >> 6) + 1u;
        uint32_t to = ((start + span + ahead) >> 6) + 1u;
        if (to > lim) to = lim;
        if (to > from + 40u) to = from + 40u;
        if (!has || span > ahead_span) to = 0u;
        typedef const uint32_t __attribute__((address_space(1)))* gp32;
        typedef uint32_t __attribute__((address_space(3)))* lp32;
        const uint8_t* bp = p.img + bbase;
        for (uint32_t sct = from; sct < to; ++sct)
          __builtin_amdgcn_global_load_lds((gp32)(bp + (size_t)sct * 64u), (lp32)ahead_sink, 4, 0, 0);
      }
    }
    // counts and places from ballots, one per bit of the piece count (one in all when no span is longer than a piece)
    const unsigned long long m_multi = __ballot(multi);
    uint32_t step_entries = 0, below_e = 0;
    if (m_multi == 0ull) {
      step_entries = (uint32_t)__popcll(m_has);
      below_e = lanes_below(m_has);
    } else {
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const unsigned long long mb = __ballot((c >> b) & 1u);
        step_entries += (uint32_t)__popcll(mb) << b;
        below_e += lanes_below(mb) << b;
      }
    }
    if (step_entries > p_open * 8u - p_count) {
      const uint32_t t = p_open + (uint32_t)lane;
      const bool mine = (uint32_t)lane < kWalStepRounds && lds_load(&seq[t % RR]) == t;
      const unsigned long long m = __ballot(mine);
      p_open += m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
      if (p_open > (p_count >> 3) + kWalStepRounds) p_open = (p_count >> 3) + kWalStepRounds;
      p_open = lead_cap(p_open);
    }
    const uint32_t room = p_open * 8u - p_count;
    // spans take room in list order; those that no longer fit stay in the list.  A span of several pieces also
    // needs its slot to be free.
    const uint32_t upto_e = below_e + c;
    const uint32_t my_span = p_spans + lanes_below(m_multi);
    bool before_busy = true;
    if (m_multi != 0ull) {
      const bool slot_busy = multi && lds_load(&slots[my_span % kWalSlots].remaining) != 0u;
      const unsigned long long m_busy = __ballot(slot_busy || (multi && my_span - p_spans >= (uint32_t)kWalSlots));
      before_busy = m_busy == 0ull || lane < __builtin_ctzll(m_busy);
    }
    const bool fits = has && upto_e <= room && before_busy;
    const unsigned long long m_fit = __ballot(fits);  // the first so many of the lanes that have an entry
    // Where the step's entries go.  A round takes as long as its longest entry, so entries that are alike should
    // share rounds: the step's entries are laid down by class - first all pieces longer than kSplitLong bytes
    // (every piece but a span's first is kWalPiece long), then the first pieces above kSplitShort bytes, then the
    // short ones - each class in lane order.  The places come from ballots over the lanes that fit.
    const uint32_t p0 = span - (c - 1u) * kWalPiece;  // the span's first piece: 1..kWalPiece bytes
    const bool cls_a = fits && p0 > kSplitLong, cls_b = fits && !cls_a && p0 > kSplitShort, cls_c = fits && !cls_a && !cls_b;
    uint32_t at0 = 0, at_rest = 0;  // where the span's first piece goes, and its other pieces (all of the first class)
    if (p.sort_classes && m_fit) {
      const uint32_t n_a = fits ? c - 1u + (cls_a ? 1u : 0u) : 0u;
      uint32_t tot_a = 0, below_a = 0;
      if ((m_fit & m_multi) == 0ull) {
        const unsigned long long ma = __ballot(cls_a);
        tot_a = (uint32_t)__popcll(ma);
        below_a = lanes_below(ma);
      } else {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          const unsigned long long mb = __ballot((n_a >> b) & 1u);
          tot_a += (uint32_t)__popcll(mb) << b;
          below_a += lanes_below(mb) << b;
        }
      }
      const unsigned long long mb = __ballot(cls_b), mc = __ballot(cls_c);
      const uint32_t tot_b = (uint32_t)__popcll(mb);
      at_rest = p_count + below_a + (cls_a ? 1u : 0u);
      at0 = cls_a ? p_count + below_a : cls_b ? p_count + tot_a + lanes_below(mb) : p_count + tot_a + tot_b + lanes_below(mc);
    } else {
      at0 = p_count + upto_e - c;
      at_rest = at0 + 1u;
    }
    if (fits) {
      const uint32_t slot = multi ? my_span % kWalSlots : kWalNoSlot;
      const uint32_t stop = start + span;
      for (uint32_t i = 0; i < c; ++i) {
        const uint32_t q = c - 1u - i;  // pieces behind piece i
        const uint32_t pe = stop - q * kWalPiece;
        const uint32_t ps = i == 0u ? start : pe - kWalPiece;
        const uint32_t at = i == 0u ? at0 : at_rest + i - 1u;
        ring[at % (RR * 8u)] =
            make_uint2(blk | (q << 19) | (i == 0u ? kWalEntryFirst : 0u) | kWalEntryValid | (slot << 26), ps | ((pe - ps) << 15));
      }
      if (multi) {
        slots[slot].acc = 0u;
        lds_store(&slots[slot].remaining, c + 1u);
      }
    }
    if (m_fit) {
      const uint32_t new_count = p_count + uni((uint32_t)__shfl((int)upto_e, 63 - __builtin_clzll(m_fit)));
      p_spans += (uint32_t)__popcll(m_fit & m_multi);
      // rounds that are complete now go out: lane i publishes round (old count / 8) + i
      const uint32_t r0 = p_count >> 3, r1 = new_count >> 3;
      if ((uint32_t)lane < r1 - r0) lds_release(&seq[(r0 + lane) % RR], r0 + lane + 1u);
      p_count = new_count;
      // The queues' places are free again (behind the reads above).  The entries taken from one queue are its
      // first so many (its lanes are l, l + A, ... in depth order, and the lanes that fit are the first so many
      // that have an entry): the lane that took the last of them says how many that makes.
      const uint32_t nl = (uint32_t)lane + (uint32_t)__popcll(m_q);
      const bool more = nl < 64u && ((m_fit >> (nl & 63u)) & 1ull) != 0ull;
      if (fits && !more) lds_store(&qctr[qi].y, eseq + 1u);
      if (lane == 0) lds_store(&ctl[3], r1);
    }
    return m_fit == m_has ? 0 : 1;
  };

  // A checksum that does not match: counted, and the record's index inside its block - found by walking the
  // block's headers again, this is rare - lowers the block's first_bad.
  auto mismatch = [&](uint32_t blk, uint32_t start) {
    atomicAdd(&p.counts[TCRC_CNT_MISMATCH], 1ull);
    atomicMin(&p.counts[TCRC_CNT_FIRST_BAD], (unsigned long long)(p.block_base + blk));
    const uint64_t base = (uint64_t)blk * kLogBlockBytes;
    const uint64_t end = base + kLogBlockBytes < p.size ? base + kLogBlockBytes : p.size;
    const bool last = p.tail && end == p.size;
    uint32_t pos = 0, k = 0, hs = 0, ln = 0, code = 0;
    while (pos + 6u < start && wal_step(p.img, base + pos, end, last, p.log_number, &hs, &ln, &code)) {
      pos += hs + ln;
      ++k;
    }
    atomicMin(&p.blocks[blk].first_bad, k);
  };
  // Walker and placer, with nothing of their own to do, take a round that is out and that nobody has claimed -
  // never one they would have to wait for, and only while enough of them are waiting that the other waves do not
  // run dry meanwhile.
  auto try_claim = [&]() -> uint32_t {
    uint32_t got = 0xffffffffu;
    if (lane == 0) {
      const uint32_t out = lds_load(&ctl[3]);
      const uint32_t need = (p.tune & 2u) ? 1u : kSplitHelpMin;
      uint32_t h = lds_load(&ctl[0]);
      while (h + need <= out &&
             !__hip_atomic_compare_exchange_strong(&ctl[0], &h, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP)) {
      }
      if (h + need <= out) got = h;
    }
    return uni(got);
  };

  // One round: the entries are copied out, the round is handed back to the placer, and the eight pieces are
  // checksummed.
  auto do_round = [&](uint32_t t) {
    const uint2 e = ring[(t % RR) * 8u + grp];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane has its entry before the round is given away
    if (lane == 0) lds_release(&seq[t % RR], t + RR);
    const bool valid = (e.x & kWalEntryValid) != 0u;
    const bool first = (e.x & kWalEntryFirst) != 0u;
    const uint32_t blk = e.x & 0x7ffffu, q = (e.x >> 19) & 31u, slot = e.x >> 26;
    const uint32_t start = e.y & 0x7fffu, len = e.y >> 15;
    const uint8_t* ptr = p.img + (uint64_t)blk * kLogBlockBytes + start;
    // the stored value is fetched now so its latency hides behind the row loop
    uint32_t stored = 0;
    if (valid && first && j == G - 1) __builtin_memcpy(&stored, ptr - 6, 4);
    const uint32_t core = span_remainder<G, NT>(lds, LB, lane, ptr, len, first ? 0u : 0xffffffffu, valid,
                                                (const uint8_t*)p.tab_main);
    // a piece with q pieces behind it enters its span's remainder multiplied by x^(8 * kWalPiece * q), on four
    // lanes of the group (see k_wal_fused)
    uint32_t shifted = core;
    if (__any(valid && q != 0u)) {
      const uint32_t rr = __shfl(core, lane | (G - 1));
      const int l = j & 3;
      uint32_t b = pow_piece[(q & 15u) * 4u + (uint32_t)l];
      uint32_t part = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        part ^= ((rr >> (31 - 8 * l - k)) & 1u) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? 0x82f63b78u : 0u);
      }
      part ^= __shfl_xor(part, 1);
      part ^= __shfl_xor(part, 2);
      if (q != 0u) shifted = part;
    }
    if (valid && j == G - 1) {
      if (slot == kWalNoSlot) {
        const uint32_t crc = ~core;
        if (stored != ((crc >> 15) | (crc << 17)) + kMaskDelta) mismatch(blk, start);
      } else {
        WalSlot* sl = &slots[slot];
        if (first) {
          lds_store(&sl->stored, stored);
          lds_store(&sl->blk, blk);
          lds_store(&sl->start, start);
        }
        (void)__hip_atomic_fetch_xor(&sl->acc, shifted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // release what this piece left in the slot, acquire what the others left
        const uint32_t before = __hip_atomic_fetch_sub(&sl->remaining, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (before == 2u) {  // the last piece: compare, then give the slot back
          const uint32_t crc = ~lds_load(&sl->acc);
          const uint32_t want = lds_load(&sl->stored);
          const uint32_t sblk = lds_load(&sl->blk), sstart = lds_load(&sl->start);
          lds_release(&sl->remaining, 0u);
          if (want != ((crc >> 15) | (crc << 17)) + kMaskDelta) mismatch(sblk, sstart);
        }
      }
    }
  };
  // Takes the next ticket and waits for its round; false when there is none any more.
  uint32_t spins = 0;
  auto next_round = [&](uint32_t& t) -> bool {
    if (lane == 0) t = atomicAdd(&ctl[0], 1u);
    t = uni(t);
    bool over = false;
    for (;;) {
      const uint32_t sq = uni(lds_acquire(&seq[t % RR]));
      if (sq == t + 1u) break;
      const uint32_t total = uni(lds_load(&ctl[1]));
      if (t >= total || lds_load(&ctl[2])) {
        over = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      if (++spins > kWalSpinLimit) {
        lds_store(&ctl[2], 1u);
        over = true;
        break;
      }
    }
    if (over) return false;
    return true;
  };
  if (wave >= 2) {
    // The fourteen waves that only checksum run a loop of their own: in one loop with the walker and the placer,
    // whatever those keep in registers across their steps (constants the compiler hoists out of the step) is live
    // across the row loops too, and a few registers more there change how the table lookups are scheduled (four
    // in flight instead of sixteen, 3 % on a scan that waits for its checksums).
    for (;;) {
      if (lds_load(&ctl[2])) break;  // some wave of the workgroup gave up waiting
      uint32_t t = 0;
      if (!next_round(t)) break;
      spins = 0;
      do_round(t);
    }
  } else {
    for (;;) {
      if (lds_load(&ctl[2])) break;  // some wave of the workgroup gave up waiting
      uint32_t t = 0;
      if (wave == 0 && !w_done) {
        if (!(p.tune & 4u)) __builtin_amdgcn_s_setprio(3);  // the walk is the critical path of a workgroup with little to do
        // the walker's cursors and counters come out of LDS (see w_save), and the header under every cursor is fetched
        {
          const uint4 sv = w_save[lane];
          w_blk = sv.x & 0x7fffffffu;
          w_pos = sv.y;
          w_k = sv.z;
          w_len = sv.w & 0xffffu;
          w_seq = sv.w >> 16;
          w_state = (sv.x >> 31) == 0u ? 0u : (w_len - w_pos >= 12u ? 3u : 1u);
          w_next = uni(u_save[0]);
        }
        WalHeader w_hdr = {0u, 0u};
        if (w_state == 3u) {
          uint2 w;
          __builtin_memcpy(&w, w_ptr() + w_pos + 4u, 8);
          w_hdr.lo = w.x;
          w_hdr.hi = w.y;
        }
        const uint32_t how = walk_run(w_hdr);
        __builtin_amdgcn_s_setprio(0);
        if (how != 1u) {
          if (how == 2u) stamp(2, 0);
          continue;
        }
        w_save[lane] = make_uint4(w_blk | (w_state != 0u ? 0x80000000u : 0u), w_pos, w_k, w_len | (w_seq << 16));
        if (lane == 0) u_save[0] = w_next;
        t = try_claim();
        if (t == 0xffffffffu) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > kWalSpinLimit) lds_store(&ctl[2], 1u);
          continue;
        }
      } else if (wave == 1 && !p_done) {
        if (!(p.tune & 4u)) __builtin_amdgcn_s_setprio(2);
        p_count = uni(u_save[5]);
        p_open = uni(u_save[6]);
        p_spans = uni(u_save[7]);
        int st = 0;
        do {
          st = place_step();
        } while (st == 0 && !p_done && !uni(lds_load(&ctl[2])));
        __builtin_amdgcn_s_setprio(0);
        if (p_done) {
          stamp(3, 64);
          continue;
        }
        if (st == 0) continue;
        if (lane == 0) {
          u_save[5] = p_count;
          u_save[6] = p_open;
          u_save[7] = p_spans;
        }
        t = try_claim();
        if (t == 0xffffffffu) {
          if (st == 2) __builtin_amdgcn_s_sleep(1);  // an empty list is looked at again soon
          else __builtin_amdgcn_s_sleep(4);
          if (++spins > kWalSpinLimit) lds_store(&ctl[2], 1u);
          continue;
        }
      } else {
        if (!next_round(t)) break;
      }
      spins = 0;
      do_round(t);
    }
  }
  stamp(31, 0);
  if (wave == 0) {
    const uint32_t total = lds_load(&ctl[6]);  // complete: wave 0 wrote it itself
    if (lane == 0) {
      if (total) atomicAdd(&p.counts[TCRC_CNT_CHECKED], (unsigned long long)total);
      if (lds_load(&ctl[2])) atomicAdd(&p.counts[TCRC_CNT_MISMATCH], 1ull << 62);  // a wait ran out: say so loudly
      if (p.records_host) {
        // the launch's record count goes to the host without a copy command (see k_wal_fused)
        unsigned long long* w = p.launch_words;
        (void)__hip_atomic_fetch_add(w, (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long before = __hip_atomic_fetch_add(w + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1ull == (unsigned long long)gridDim.x) {
          const unsigned long long sum = __hip_atomic_exchange(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          (void)__hip_atomic_exchange(w + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *p.records_host += sum;
        }
      }
    }
  }
}

// ---- log framing ---------------------------------------------------------------------------------------
// One launch writes a log image the way log::Writer::AddRecord / EmitPhysicalRecord append records
// (db/log_writer.cc:45-151): every payload byte is read once, stored to its place in the image and folded into its
// fragment's checksum from the same registers; the header (length, type, log number, and the zero bytes of a
// block tail in front of it) is written by the lane group of the fragment's first piece, the masked checksum by
// the group of its last.  The fragmentation itself - where every header goes - is the host's plan
// (WalFrameEntry): it depends on nothing but the record lengths.
// A wave takes rounds of eight entries.  The pieces of one fragment are neighbouring entries of one round; their
// remainders, multiplied by x^(8 * kFramePiece * pieces behind), are XORed together with a segmented scan over the
// wave's eight groups.
template <bool NT, bool PHASE_DST>
__global__ __launch_bounds__(kThreads) void k_wal_frame(WalFrameParams p) {
  constexpr int G = 8;
  constexpr int kAuxDwords = AuxDwords<G>::value;
  __shared__ __align__(16) uint32_t lds[kMainDwords + kAuxDwords];
  __shared__ uint32_t pow_piece[64];
  __shared__ uint32_t prefix_reg[16];  // the register after the header bytes the checksum covers, by record type

  const int tid = threadIdx.x;
  fill_main_tables(lds, p.tab_main, tid);
  for (int q = tid; q < kAuxDwords / 4; q += kThreads)
    reinterpret_cast<uint4*>(lds + kMainDwords)[q] = reinterpret_cast<const uint4*>(p.tab_aux)[q];
  if (tid < 64) pow_piece[tid] = p.pow_piece[tid];
  __syncthreads();
  const uint32_t* aux = lds + kMainDwords;
  if (tid < 16) {
    // crc = Extend(type_crc_[t], log number) (db/log_writer.cc:115,130-131): the type byte, then for the 11-byte
    // header the log number's four bytes
    const uint32_t* T0 = aux + 768;
    uint32_t reg = 0xffffffffu;
    reg = T0[(reg ^ (uint32_t)tid) & 0xffu] ^ (reg >> 8);
    if (p.header_size == kLogRecyclableHeader) {
#pragma unroll
      for (int k = 0; k < 4; ++k) reg = T0[(reg ^ (p.log_number >> (8 * k))) & 0xffu] ^ (reg >> 8);
    }
    prefix_reg[tid] = reg;
  }
  __syncthreads();

  const int lane = tid & 63;
  const int j = lane & (G - 1);
  const int grp = lane / G;
  const uint32_t LB = (uint32_t)(lane & 31) * 4u | 0x10000u;
  const uint32_t hs = p.header_size;
  // rounds are dealt to workgroups first, then to the waves of a workgroup (a small plan spreads over all CUs)
  const uint64_t wave = (uint64_t)blockIdx.x + (uint64_t)gridDim.x * (tid >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerWg;
  const uint4* entries = reinterpret_cast<const uint4*>(p.entries);
  uint4 nxt = make_uint4(0u, 0u, 0u, 0u);
  if (wave < p.nrounds) nxt = entries[wave * 8u + grp];
  for (uint64_t round = wave; round < p.nrounds; round += nwaves) {
    const uint4 cur = nxt;
    if (round + nwaves < p.nrounds) nxt = entries[(round + nwaves) * 8u + grp];  // on its way while this round runs
    const uint64_t wa = (uint64_t)cur.x | ((uint64_t)cur.y << 32), wb = (uint64_t)cur.z | ((uint64_t)cur.w << 32);
    const uint32_t plen = (uint32_t)(wa >> 40) & 0x1fffu, q = (uint32_t)(wa >> 53) & 7u;
    const bool first = (wa >> 56) & 1u, last = (wa >> 57) & 1u, valid = (wa >> 58) & 1u;
    const uint32_t pad = (uint32_t)(wa >> 59) & 15u;
    const uint32_t flen = (uint32_t)(wb >> 40) & 0xffffu, type = (uint32_t)(wb >> 56) & 15u;
    const uint8_t* ptr = p.src + (wa & kFrameOffMask);
    uint8_t* out = p.dst + (wb & kFrameOffMask);
    // span_remainder takes a finalised value to continue from: the first piece continues the header bytes,
    // the others start from a zero register
    const uint32_t init = first ? ~prefix_reg[type] : 0xffffffffu;
    const uint32_t core = span_remainder<G, NT, true, PHASE_DST>(lds, LB, lane, ptr, plen, init, valid,
                                                                  (const uint8_t*)p.tab_main, out, p.sink);
    // header bytes 4 .. hs-1 (length, type, log number) and the zero bytes in front of the header: one byte per lane
    if (valid && first) {
      uint8_t* h = out - hs;
      const uint32_t bi = 4u + (uint32_t)j;
      if (bi < hs) {
        const uint32_t val = bi == 4u ? flen : bi == 5u ? flen >> 8 : bi == 6u ? type : p.log_number >> (8u * (bi - 7u));
        h[bi] = (uint8_t)val;
      }
      if ((uint32_t)j < pad) h[-1 - j] = 0;
      if ((uint32_t)j + 8u < pad) h[-9 - j] = 0;
    }
    // the piece's remainder times x^(8 * kFramePiece * q), on four lanes of the group (see k_wal_fused)
    uint32_t shifted = core;
    if (__any(valid && q != 0u)) {
      const uint32_t rr = __shfl(core, lane | (G - 1));
      const int l = j & 3;
      uint32_t b = pow_piece[q * (kFramePiece / kWalPiece) * 4u + (uint32_t)l];
      uint32_t part = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        part ^= ((rr >> (31 - 8 * l - k)) & 1u) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? 0x82f63b78u : 0u);
      }
      part ^= __shfl_xor(part, 1);
      part ^= __shfl_xor(part, 2);
      if (q != 0u) shifted = part;
    }
    // XOR over the pieces of each fragment: a segmented scan over the eight groups (a fragment's pieces are
    // neighbours, all in this round)
    uint32_t v = valid ? __shfl(shifted, lane | (G - 1)) : 0u;
    if (__any(valid && !first)) {
      int head = (first || !valid) ? 1 : 0;
#pragma unroll
      for (int d = 1; d < 8; d <<= 1) {
        const uint32_t vu = __shfl_up(v, 8 * d);
        const int hu = __shfl_up(head, 8 * d);
        if (grp >= d && !head) {
          v ^= vu;
          head = hu;
        }
      }
    }
    if (valid && last && j == G - 1) {
      const uint32_t crc = ~v;
      const uint32_t masked = ((crc >> 15) | (crc << 17)) + kMaskDelta;
      uint8_t* h = out - (flen - plen) - hs;  // the fragment's header: its payload begins flen - plen bytes before this piece
      __builtin_memcpy(h, &masked, 4);
    }
  }
}

// Streaming-read probe: what a kernel that only loads (16 B per lane, whole lines, XOR-folded)
// reaches on this device; the ceiling the CRC kernels are compared with.
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_probe_read(const uint8_t* __restrict__ base, uint64_t n16, uint32_t* out) {
  const uint64_t nthreads = (uint64_t)gridDim.x * kThreads;
  uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
#define SRC(I_) load16<NT>(base + 16 * (I_))
  for (; i + 3 * nthreads < n16; i += 4 * nthreads) {
    const uint4 a = SRC(i), b = SRC(i + nthreads), c = SRC(i + 2 * nthreads), d = SRC(i + 3 * nthreads);
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x;
    acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
    acc.z ^= a.z ^ b.z ^ c.z ^ d.z;
    acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
  }
  for (; i < n16; i += nthreads) {
    const uint4 a = SRC(i);
    acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
  }
#undef SRC
  out[(uint64_t)blockIdx.x * kThreads + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

__global__ void k_counts_reset(unsigned long long* c) {
  c[TCRC_CNT_MISMATCH] = 0ull;
  c[TCRC_CNT_FIRST_BAD] = ~0ull;
  c[TCRC_CNT_CHECKED] = 0ull;
}

constexpr int kGeoLanes[kGeoCount] = {4, 8, 16, 64};


int geo_index(int g) {
  for (int i = 0; i < kGeoCount; ++i)
    if (kGeoLanes[i] == g) return i;
  return -1;
}

template <int G>
int launch_batch(tcrc_ctx* ctx, BatchParams p, hipStream_t s) {
  const int gi = geo_index(G);
  p.tab_main = ctx->d_tab_main[gi];
  p.tab_aux = ctx->d_tab_aux[gi];
  const uint64_t rounds = (p.n + (64 / G) - 1) / (64 / G);
  uint64_t wgs = rounds;
  if (wgs > (uint64_t)ctx->cus) wgs = (uint64_t)ctx->cus;
  if (wgs == 0) wgs = 1;
  const bool in_image = (p.flags & (TCRC_F_STORE_AFTER | kFlagStampBefore)) != 0;
  if (ctx->nontemporal) {
    if (in_image)
      hipLaunchKernelGGL((k_crc_batch<G, true, kModeInImage>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p, NoLongArgs{});
    else
      hipLaunchKernelGGL((k_crc_batch<G, true, kModePlain>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p, NoLongArgs{});
  } else {
    if (in_image)
      hipLaunchKernelGGL((k_crc_batch<G, false, kModeInImage>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p, NoLongArgs{});
    else
      hipLaunchKernelGGL((k_crc_batch<G, false, kModePlain>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p, NoLongArgs{});
  }
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}

}  // namespace

namespace tcrc_internal {

// The lookup sets one geometry (G lanes per buffer, rows of D = 16 G bytes) keeps in LDS:
//   main: Z_D as four 256-entry tables (replicated 32x when copied into LDS);
//   aux:  Z_4, then Z_16, Z_32, ... Z_(8 G) for the lane reduction (1 + log2 G sets of 1024 words),
//         then x^(-8 t) * x^(32 l / UL) for t < D and l < UL, the multipliers that take t trailing zero
//         bytes back out, pre-shifted for each of the UL lanes that share the multiplication.
void build_tables(int G, std::vector<uint32_t>* main_tab, std::vector<uint32_t>* aux) {
  int log2g = 0;
  while ((1 << log2g) < G) ++log2g;
  const int ul = G <= 16 ? 4 : 1;
  uint32_t t[4][256];
  tcrc::build_shift_tables(16u * G, t);
  main_tab->assign(&t[0][0], &t[0][0] + 1024);
  aux->assign((size_t)(1 + log2g) * 1024 + (size_t)16 * G * ul, 0u);
  tcrc::build_shift_tables(4, t);
  memcpy(aux->data(), t, sizeof(t));
  for (int st = 1; st <= log2g; ++st) {
    tcrc::build_shift_tables(16u << (st - 1), t);
    memcpy(aux->data() + st * 1024, t, sizeof(t));
  }
  for (int k = 0; k < 16 * G; ++k)
    for (int l = 0; l < ul; ++l)  // lane l multiplies bits [32 l / ul, 32 (l+1) / ul) of the register (from x^0 up)
      (*aux)[(size_t)(1 + log2g) * 1024 + k * ul + l] =
          tcrc::gf2_mulmod(tcrc::gf2_xinv_bytes((uint32_t)k), tcrc::gf2_xpow(32u * l / ul));
}

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

int scratch_reserve(tcrc_ctx* ctx, tcrc_scratch* s, size_t bytes) {
  if (s->bytes >= bytes) return TCRC_OK;
  TCRC_DEVICE_SCOPE(ctx);
  if (s->p) {
    HIP_TRY(hipDeviceSynchronize());  // earlier work may still use the old block
    (void)hipFree(s->p);
    s->p = nullptr;
    s->bytes = 0;
  }
  size_t want = bytes + bytes / 4 + 4096;
  HIP_TRY(hipMalloc(&s->p, want));
  s->bytes = want;
  return TCRC_OK;
}

void scratch_release(tcrc_scratch* s) {
  if (s->p) (void)hipFree(s->p);
  s->p = nullptr;
  s->bytes = 0;
}

int run_segments(tcrc_ctx* ctx, BatchParams p, const LongArgs& la, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (p.n == 0) return TCRC_OK;
  if (!p.base) return fail(TCRC_ERR_INVALID, "null base pointer");
  if (la.nbuf == 0 || la.nbuf > (uint32_t)kLongLdsBuffers || (!la.bufs && la.nbuf > (uint32_t)kLongInline))
    return fail(TCRC_ERR_INVALID, "segment launch: buffer count out of range");
  // eight segments per wave are folded before one atomic: wider groups would queue on the accumulator
  constexpr int G = 8;
  const int gi = geo_index(G);
  p.tab_main = ctx->d_tab_main[gi];
  p.tab_aux = ctx->d_tab_aux[gi];
  p.flags = kFlagSegments;
  p.uniform_suffix = -1;
  const uint64_t rounds = (p.n + (64 / G) - 1) / (64 / G);
  // A small launch spreads over 64 workgroups first (a CU alone pulls its bytes slowly), then gives each of
  // them up to eight waves of work, then takes more workgroups, and only with every CU busy goes on to
  // sixteen waves: every workgroup ends with one arrival at an accumulator line, and arrivals at one line
  // queue (about 20 ns each); sixteen waves of a few rows each on one CU cost more (every segment has some
  // 700 instructions of edge work) than the same rows on two CUs (16 MiB in one buffer: 26.5 us with sixteen
  // waves first, 23.7 us with eight).
  constexpr uint64_t kSegFill = 8;
  uint64_t wgs64 = rounds < 64 ? rounds : 64;
  const uint64_t filled = (rounds + kSegFill - 1) / kSegFill;
  if (filled > wgs64) wgs64 = filled;
  const unsigned wgs = (unsigned)(wgs64 < (uint64_t)ctx->cus ? wgs64 : (uint64_t)ctx->cus);
  ctx->st_batch_launches.fetch_add(1, std::memory_order_relaxed);
  ctx->st_batch_buffers.fetch_add(p.n, std::memory_order_relaxed);
  hipStream_t s = (hipStream_t)stream;
  if (ctx->nontemporal)
    hipLaunchKernelGGL((k_crc_batch<G, true, kModeSegments>), dim3(wgs), dim3(kThreads), 0, s, p, la);
  else
    hipLaunchKernelGGL((k_crc_batch<G, false, kModeSegments>), dim3(wgs), dim3(kThreads), 0, s, p, la);
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}

int run_wal_scan(tcrc_ctx* ctx, WalScanParams p, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (p.nblocks == 0) return TCRC_OK;
  if (p.nblocks > kWalMaxBlocks) return fail(TCRC_ERR_TOO_LARGE, "log slice above 16 GiB");
  if (!p.img || !p.blocks || !p.counts || (p.records_host && !p.launch_words))
    return fail(TCRC_ERR_INVALID, "null argument");
  const int gi = geo_index(8);
  p.tab_main = ctx->d_tab_main[gi];
  p.tab_aux = ctx->d_tab_aux[gi];
  memcpy(p.pow_piece, wal_piece_powers(), sizeof(p.pow_piece));
  p.trace = (unsigned long long*)ctx->wal_trace;
  p.sort_classes = ctx->wal_sort ? 1u : 0u;
  p.tune = ctx->wal_tune;
  // chunks of up to 64 blocks (one per walker lane), at least two chunks per workgroup
  p.chunk_shift = 6;
  while (p.chunk_shift > 0 && (p.nblocks >> p.chunk_shift) < 2u * (uint32_t)ctx->cus) --p.chunk_shift;
  if (ctx->forced_wal_chunk_shift >= 0) p.chunk_shift = (uint32_t)ctx->forced_wal_chunk_shift;
  const unsigned wgs = p.nblocks < (uint32_t)ctx->cus ? p.nblocks : (unsigned)ctx->cus;
  hipStream_t s = (hipStream_t)stream;
  if (ctx->wal_split) {
    if (p.trace)
      hipLaunchKernelGGL((k_wal_split<true, true>), dim3(wgs), dim3(kThreads), 0, s, p);
    else if (ctx->nontemporal)
      hipLaunchKernelGGL((k_wal_split<true, false>), dim3(wgs), dim3(kThreads), 0, s, p);
    else
      hipLaunchKernelGGL((k_wal_split<false, false>), dim3(wgs), dim3(kThreads), 0, s, p);
  } else if (ctx->nontemporal) {
    hipLaunchKernelGGL(k_wal_fused<true>, dim3(wgs), dim3(kThreads), 0, s, p);
  } else {
    hipLaunchKernelGGL(k_wal_fused<false>, dim3(wgs), dim3(kThreads), 0, s, p);
  }
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}


const uint32_t* wal_piece_powers() {
  // worked out once per process (64 modular powers take tens of microseconds on the host, as long as the scan of
  // a small log takes on the device)
  struct PiecePowers {
    uint32_t v[64];
    PiecePowers() {
      for (uint32_t q = 0; q < 16; ++q)
        for (uint32_t l = 0; l < 4; ++l) v[q * 4 + l] = tcrc::gf2_xpow(8ull * kWalPiece * q + 8ull * l);
    }
  };
  static const PiecePowers piece_powers;
  return piece_powers.v;
}

int run_wal_frame(tcrc_ctx* ctx, WalFrameParams p, bool phase_dst, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (p.nrounds == 0) return TCRC_OK;
  if (!p.src || !p.dst || !p.entries) return fail(TCRC_ERR_INVALID, "null argument");
  if (!ctx->d_sink) return fail(TCRC_ERR_INVALID, "context without a store sink");
  const int gi = geo_index(8);
  p.tab_main = ctx->d_tab_main[gi];
  p.tab_aux = ctx->d_tab_aux[gi];
  p.sink = ctx->d_sink;
  memcpy(p.pow_piece, wal_piece_powers(), sizeof(p.pow_piece));
  const uint64_t wgs = p.nrounds < (uint64_t)ctx->cus ? p.nrounds : (uint64_t)ctx->cus;
  hipStream_t s = (hipStream_t)stream;
  if (ctx->nontemporal) {
    if (phase_dst) hipLaunchKernelGGL((k_wal_frame<true, true>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((k_wal_frame<true, false>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p);
  } else {
    if (phase_dst) hipLaunchKernelGGL((k_wal_frame<false, true>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((k_wal_frame<false, false>), dim3((unsigned)wgs), dim3(kThreads), 0, s, p);
  }
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}

int run_batch(tcrc_ctx* ctx, BatchParams p, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (p.n == 0) return TCRC_OK;
  if (!p.base) return fail(TCRC_ERR_INVALID, "null base pointer");
  TCRC_DEVICE_SCOPE(ctx);
  hipStream_t s = (hipStream_t)stream;
  int g = ctx->forced_g;  // set for experiments only
  if (g == 0 && p.lanes) g = p.lanes;  // the caller knows the shape of its spans (log scans)
  if (g == 0) {
    g = 8;
    // A batch too small to give every wave of the chip eight buffers is latency-bound: spread each
    // buffer over more lanes instead (a whole wave per buffer when there are fewer buffers than waves).
    const uint64_t waves = (uint64_t)ctx->cus * 16;
    if (!p.n_dev) {
      if (p.n <= waves) g = 64;
      else if (p.n <= waves * 4) g = 16;
    }
    // Short buffers in numbers: four lanes each.  What a wave spends per buffer outside the row loop is then
    // shared by sixteen buffers; above about 1.4 KiB eight lanes win again, because only they ask for whole
    // 128-byte lines in one instruction (four lanes stop at 5.2-5.5 TB/s whatever the length).
    constexpr uint32_t kShortSpan = 1408;
    const bool short_spans = p.desc ? (p.flags & TCRC_F_SHORT_SPANS) != 0 : p.block_len <= kShortSpan;
    if (g == 8 && short_spans) g = 4;
  }
  ctx->st_batch_launches.fetch_add(1, std::memory_order_relaxed);
  ctx->st_batch_buffers.fetch_add(p.n, std::memory_order_relaxed);
  switch (g) {
    case 4: return launch_batch<4>(ctx, p, s);
    case 8: return launch_batch<8>(ctx, p, s);
    case 16: return launch_batch<16>(ctx, p, s);
    case 64: return launch_batch<64>(ctx, p, s);
    default: return fail(TCRC_ERR_INVALID, "unsupported lanes-per-buffer setting");
  }
}

}  // namespace tcrc_internal

extern "C" {

const char* tcrc_last_error(void) { return tcrc_internal::g_err.c_str(); }

uint32_t tcrc_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + kMaskDelta; }
uint32_t tcrc_unmask(uint32_t masked) {
  const uint32_t rot = masked - kMaskDelta;
  return (rot >> 17) | (rot << 15);
}

int tcrc_init(int device, tcrc_ctx** out) {
  if (!out) return fail(TCRC_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int count = 0;
  const hipError_t ce = hipGetDeviceCount(&count);
  if (ce != hipSuccess || count <= 0)
    return fail(TCRC_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(ce) +
                                        "): the tcrc engine has no CPU fallback");
  if (device < 0 || device >= count) return fail(TCRC_ERR_INVALID, "device index out of range");
  tcrc_internal::DeviceScope device_scope(device);
  HIP_TRY(device_scope.err);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  tcrc_ctx* ctx = new tcrc_ctx();
  ctx->device = device;
  ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("TCRC_LANES_PER_BUFFER")) ctx->forced_g = atoi(e);
  if (const char* e = getenv("TCRC_WAL_CHUNK_SHIFT")) {
    const int v = atoi(e);
    if (v >= 0 && v <= 20) ctx->forced_wal_chunk_shift = v;
  }
  if (const char* e = getenv("TCRC_NONTEMPORAL")) ctx->nontemporal = atoi(e) != 0;
  if (const char* e = getenv("TCRC_NO_POLL")) ctx->poll_completion = atoi(e) == 0;
  if (const char* e = getenv("TCRC_DEBUG_NOALLOC_MEMSET")) ctx->debug_noalloc_memset = atoi(e) != 0;
  for (int gi = 0; gi < kGeoCount; ++gi) {
    const int G = kGeoLanes[gi];
    int log2g = 0;
    while ((1 << log2g) < G) ++log2g;
    std::vector<uint32_t> main_tab, aux;
    build_tables(G, &main_tab, &aux);
    hipError_t e = hipMalloc(&ctx->d_tab_main[gi], main_tab.size() * 4);
    if (e == hipSuccess)
      e = hipMemcpy(ctx->d_tab_main[gi], main_tab.data(), main_tab.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_tab_aux[gi], aux.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tab_aux[gi], aux.data(), aux.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      tcrc_shutdown(ctx);
      return fail(TCRC_ERR_HIP, std::string("table upload: ") + hipGetErrorString(e));
    }
  }
  {
    hipError_t e = hipMalloc((void**)&ctx->d_sink, 4096);
    if (e != hipSuccess) {
      tcrc_shutdown(ctx);
      return fail(TCRC_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
  }
  if (const char* e = getenv("TCRC_WAL_SPLIT")) ctx->wal_split = atoi(e) != 0;
  if (const char* e = getenv("TCRC_WAL_SORT")) ctx->wal_sort = atoi(e) != 0;
  if (const char* e = getenv("TCRC_WAL_TUNE")) ctx->wal_tune = (uint32_t)atoi(e);
  if (const char* e = getenv("TCRC_FRAME_PHASE")) ctx->frame_phase_dst = atoi(e) != 0;
  if (int rc = long_device_setup(ctx)) {
    tcrc_shutdown(ctx);
    return rc;
  }
  char buf[256];
  snprintf(buf, sizeof(buf), "Supported on %s (%s, %d CUs): LDS-table batched CRC32C", prop.name, prop.gcnArchName,
           ctx->cus);
  ctx->description = buf;
  *out = ctx;
  return TCRC_OK;
}

int tcrc_debug_tables(int lanes, uint32_t* main_out, uint32_t* aux_out, size_t aux_capacity, size_t* aux_words) {
  if (geo_index(lanes) < 0) return fail(TCRC_ERR_INVALID, "lanes must be 4, 8, 16 or 64");
  std::vector<uint32_t> main_tab, aux;
  build_tables(lanes, &main_tab, &aux);
  if (aux_words) *aux_words = aux.size();
  if (main_out) memcpy(main_out, main_tab.data(), 4096);
  if (aux_out) {
    if (aux_capacity < aux.size()) return fail(TCRC_ERR_INVALID, "aux_capacity too small");
    memcpy(aux_out, aux.data(), aux.size() * 4);
  }
  return TCRC_OK;
}

int tcrc_shutdown(tcrc_ctx* ctx) {
  if (!ctx) return TCRC_OK;
  tcrc_internal::DeviceScope device_scope(ctx->device);
  long_device_release(ctx);
  for (int gi = 0; gi < kGeoCount; ++gi) {
    if (ctx->d_tab_main[gi]) (void)hipFree(ctx->d_tab_main[gi]);
    if (ctx->d_tab_aux[gi]) (void)hipFree(ctx->d_tab_aux[gi]);
  }
  for (tcrc_scratch* s : {&ctx->wal_misc, &ctx->wal_list_counts,
                          &ctx->wal_list_first, &ctx->wal_list_temp, &ctx->wal_frags,
                          &ctx->long_in, &ctx->long_acc})
    scratch_release(s);
  for (tcrc_lane* l : ctx->lanes) {
    for (tcrc_scratch* s : {&l->data, &l->desc, &l->out, &l->aux, &l->blocks, &l->records}) scratch_release(s);
    if (l->stream) (void)hipStreamDestroy(l->stream);
    if (l->zc) (void)hipHostFree(l->zc);
    if (l->done_counter) (void)hipFree(l->done_counter);
    delete l;
  }
  ctx->lanes.clear();
  if (ctx->d_sink) (void)hipFree(ctx->d_sink);
  for (auto& fb : ctx->frame_free) {
    if (fb.h) (void)hipHostFree(fb.h);
    if (fb.d) (void)hipFree(fb.d);
  }
  if (ctx->frame_stream) (void)hipStreamDestroy(ctx->frame_stream);
  for (auto& e : ctx->pin_free) (void)hipHostFree(e.first);
  stage_pool_destroy(ctx);
  if (ctx->long_pin) (void)hipHostFree(ctx->long_pin);
  if (ctx->long_flag) (void)hipHostFree(ctx->long_flag);
  if (ctx->wal_pin) (void)hipHostFree(ctx->wal_pin);
  delete ctx;
  return TCRC_OK;
}

int tcrc_get_stats(tcrc_ctx* ctx, tcrc_stats* out) {
  if (!ctx || !out) return fail(TCRC_ERR_INVALID, "null argument");
  out->batch_launches = ctx->st_batch_launches.load();
  out->batch_buffers = ctx->st_batch_buffers.load();
  out->host_calls = ctx->st_host_calls.load();
  out->host_bytes_in = ctx->st_host_bytes_in.load();
  out->host_nanos = ctx->st_host_nanos.load();
  out->log_images = ctx->st_log_images.load();
  out->log_bytes = ctx->st_log_bytes.load();
  out->long_calls = ctx->st_long_calls.load();
  out->long_bytes = ctx->st_long_bytes.load();
  {
    std::lock_guard<std::mutex> lock(ctx->lanes_mu);
    out->host_lanes = ctx->lanes.size();
  }
  return TCRC_OK;
}

int tcrc_reset_stats(tcrc_ctx* ctx) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  for (std::atomic<uint64_t>* a : {&ctx->st_batch_launches, &ctx->st_batch_buffers, &ctx->st_host_calls,
                                   &ctx->st_host_bytes_in, &ctx->st_host_nanos, &ctx->st_log_images, &ctx->st_log_bytes,
                                   &ctx->st_long_calls, &ctx->st_long_bytes})
    a->store(0);
  return TCRC_OK;
}

const char* tcrc_describe(tcrc_ctx* ctx) { return ctx ? ctx->description.c_str() : "no context"; }

int tcrc_dev_counts_reset(tcrc_ctx* ctx, uint64_t* d_counts, void* stream) {
  if (!ctx || !d_counts) return fail(TCRC_ERR_INVALID, "null argument");
  TCRC_DEVICE_SCOPE(ctx);
  hipLaunchKernelGGL(k_counts_reset, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)d_counts);
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}

int tcrc_dev_probe_read(tcrc_ctx* ctx, const void* d_base, uint64_t nbytes, uint32_t* d_scratch, uint32_t wgs_per_cu,
                        void* stream) {
  if (!ctx || !d_base || !d_scratch) return fail(TCRC_ERR_INVALID, "null argument");
  if (((uintptr_t)d_base & 15u) != 0) return fail(TCRC_ERR_INVALID, "probe needs a 16-byte aligned base");
  if (wgs_per_cu == 0 || wgs_per_cu > 2) return fail(TCRC_ERR_INVALID, "wgs_per_cu must be 1 or 2");
  TCRC_DEVICE_SCOPE(ctx);
  if (ctx->nontemporal)
    hipLaunchKernelGGL(k_probe_read<true>, dim3((unsigned)ctx->cus * wgs_per_cu), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint8_t*)d_base, nbytes / 16, d_scratch);
  else
    hipLaunchKernelGGL(k_probe_read<false>, dim3((unsigned)ctx->cus * wgs_per_cu), dim3(kThreads), 0,
                       (hipStream_t)stream, (const uint8_t*)d_base, nbytes / 16, d_scratch);
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}

// Only the documented flag bits may come in through the C ABI (the higher ones are internal).
static int check_flags(uint32_t flags, uint32_t allowed) {
  if (flags & ~allowed) return fail(TCRC_ERR_INVALID, "unknown or misplaced flag bits");
  return TCRC_OK;
}
constexpr uint32_t kStampFlags = TCRC_F_MASK | TCRC_F_STORE_AFTER | TCRC_F_SHORT_SPANS;
constexpr uint32_t kVerifyFlags = TCRC_F_MASK | TCRC_F_EXPECT_AFTER | TCRC_F_EXPECT_BEFORE | TCRC_F_SHORT_SPANS;

static int check_uniform(uint64_t n_blocks, uint32_t block_len, uint64_t stride) {
  if (block_len > (1u << 30)) return fail(TCRC_ERR_TOO_LARGE, "block_len above 1 GiB: use the long-buffer entry point");
  if (n_blocks > 1 && stride < block_len) return fail(TCRC_ERR_INVALID, "stride smaller than block_len");
  return TCRC_OK;
}

int tcrc_dev_blocks_stamp(tcrc_ctx* ctx, const void* d_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                          uint32_t init, int suffix_byte, const uint8_t* d_suffix, uint32_t flags, uint32_t* d_out,
                          void* stream) {
  if (int rc = check_uniform(n_blocks, block_len, stride)) return rc;
  if (int rc = check_flags(flags, kStampFlags)) return rc;
  if (!d_out && !(flags & TCRC_F_STORE_AFTER) && n_blocks) return fail(TCRC_ERR_INVALID, "null output pointer");
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.n = n_blocks;
  p.stride = stride;
  p.block_len = block_len;
  p.uniform_init = init;
  p.suffix = d_suffix;
  p.uniform_suffix = suffix_byte >= 0 ? (suffix_byte & 0xff) : -1;
  p.flags = flags;
  p.out = d_out;
  return run_batch(ctx, p, stream);
}

int tcrc_dev_blocks_verify(tcrc_ctx* ctx, const void* d_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                           uint32_t init, int suffix_byte, const uint8_t* d_suffix, uint32_t flags,
                           const uint32_t* d_expected, uint32_t* d_out, uint8_t* d_ok, uint64_t* d_counts,
                           void* stream) {
  if (int rc = check_uniform(n_blocks, block_len, stride)) return rc;
  if (int rc = check_flags(flags, kVerifyFlags)) return rc;
  if (!d_counts) return fail(TCRC_ERR_INVALID, "verify needs a counts block");
  if (!d_expected && !(flags & (TCRC_F_EXPECT_AFTER | TCRC_F_EXPECT_BEFORE)))
    return fail(TCRC_ERR_INVALID, "verify needs d_expected or an in-image expectation flag");
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.n = n_blocks;
  p.stride = stride;
  p.block_len = block_len;
  p.uniform_init = init;
  p.suffix = d_suffix;
  p.uniform_suffix = suffix_byte >= 0 ? (suffix_byte & 0xff) : -1;
  p.flags = flags | TCRC_F_MASK;
  p.verify = 1;
  p.out = d_out;
  p.expected = d_expected;
  p.ok = d_ok;
  p.counts = (unsigned long long*)d_counts;
  return run_batch(ctx, p, stream);
}

int tcrc_dev_batch_stamp(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n,
                         const uint8_t* d_suffix, uint32_t flags, uint32_t* d_out, void* stream) {
  return tcrc_internal::batch_stamp(ctx, d_base, d_desc, n, d_suffix, flags, d_out, stream, DoneSignal{});
}

int tcrc_dev_batch_verify(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n,
                          const uint8_t* d_suffix, uint32_t flags, const uint32_t* d_expected, uint32_t* d_out,
                          uint8_t* d_ok, uint64_t* d_counts, void* stream) {
  return tcrc_internal::batch_verify(ctx, d_base, d_desc, n, d_suffix, flags, d_expected, d_out, d_ok, d_counts, stream,
                                     DoneSignal{});
}

}  // extern "C"

namespace tcrc_internal {

int batch_stamp(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n, const uint8_t* d_suffix,
                uint32_t flags, uint32_t* d_out, void* stream, const DoneSignal& done) {
  if (int rc = check_flags(flags, kStampFlags)) return rc;
  if (n && (!d_desc || (!d_out && !(flags & TCRC_F_STORE_AFTER))))
    return fail(TCRC_ERR_INVALID, "null descriptor or output pointer");
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.desc = d_desc;
  p.n = n;
  p.suffix = d_suffix;
  p.uniform_suffix = -1;
  p.flags = flags;
  p.out = d_out;
  p.done = done;
  return run_batch(ctx, p, stream);
}

int batch_verify(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n, const uint8_t* d_suffix,
                 uint32_t flags, const uint32_t* d_expected, uint32_t* d_out, uint8_t* d_ok, uint64_t* d_counts,
                 void* stream, const DoneSignal& done) {
  if (n && !d_desc) return fail(TCRC_ERR_INVALID, "null descriptor pointer");
  if (int rc = check_flags(flags, kVerifyFlags)) return rc;
  if (!d_counts) return fail(TCRC_ERR_INVALID, "verify needs a counts block");
  if (!d_expected && !(flags & (TCRC_F_EXPECT_AFTER | TCRC_F_EXPECT_BEFORE)))
    return fail(TCRC_ERR_INVALID, "verify needs d_expected or an in-image expectation flag");
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.desc = d_desc;
  p.n = n;
  p.suffix = d_suffix;
  p.uniform_suffix = -1;
  p.flags = flags | TCRC_F_MASK;
  p.verify = 1;
  p.out = d_out;
  p.expected = d_expected;
  p.ok = d_ok;
  p.counts = (unsigned long long*)d_counts;
  p.done = done;
  return run_batch(ctx, p, stream);
}

}  // namespace tcrc_internal
