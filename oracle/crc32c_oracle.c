/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the reference's CRC-32C ("Castagnoli") arithmetic, used
 * only as the checker in tests/, in the smoke test and as the timed CPU
 * baseline of the benchmark.  Nothing under terarkdb_amd/ links, loads or
 * calls this file.
 *
 * What it restates (paths relative to the reference tree):
 *   util/crc32c.h:23-29      Extend(init, data, n), Value(data, n) = Extend(0, ..)
 *   util/crc32c.h:31-47      Mask / Unmask (rotate by 15, add 0xa282ead8)
 *   util/crc32c.cc:356-398   ExtendImpl: l = crc ^ 0xffffffff; byte steps
 *                            l = table0[(l ^ b) & 0xff] ^ (l >> 8); return l ^ 0xffffffff
 *   util/crc32c.cc:326-340   Slow_CRC32: slice-by-4 over little-endian words
 *   util/crc32c.cc:342-354   Fast_CRC32: SSE4.2 crc32 instruction, 8 bytes at a time
 *   util/crc32c.cc:668-1199  crc32c_3way: three interleaved crc32 streams merged
 *                            with carry-less multiplies (restated here as
 *                            oracle_crc32c_extend_hw with constants derived at
 *                            start-up instead of a literal table)
 *
 * The reference's generator polynomial is the reflected 0x82f63b78
 * (table0_[0x80], util/crc32c.cc:83).  The tables are generated from it rather
 * than transcribed.
 *
 * Parity status: pinned.  tests/test_oracle_crc.py checks every function here
 * against the known answers the reference's own test holds
 * (util/crc32c_test.cc:31-94) and against values captured from a build of the
 * reference (tests/golden/reference_captured.json).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <nmmintrin.h>
#include <wmmintrin.h>
#define ORACLE_X86 1
#else
#define ORACLE_X86 0
#endif

#define ORACLE_POLY_REFLECTED 0x82f63b78u

static uint32_t g_tab[4][256];
static int g_tab_ready = 0;

static void oracle_build_tables(void) {
  if (g_tab_ready) return;
  for (uint32_t b = 0; b < 256; b++) {
    uint32_t r = b;
    for (int k = 0; k < 8; k++) r = (r & 1) ? (r >> 1) ^ ORACLE_POLY_REFLECTED : (r >> 1);
    g_tab[0][b] = r;
  }
  /* g_tab[k][b] = register after feeding byte b followed by k zero bytes. */
  for (int k = 1; k < 4; k++)
    for (uint32_t b = 0; b < 256; b++) {
      uint32_t r = g_tab[k - 1][b];
      g_tab[k][b] = g_tab[0][r & 0xff] ^ (r >> 8);
    }
  g_tab_ready = 1;
}

/* Definition-level CRC: one bit at a time (used to pin the table paths). */
uint32_t oracle_crc32c_extend_bitwise(uint32_t init, const void* data, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
  uint32_t l = init ^ 0xffffffffu;
  for (size_t i = 0; i < n; i++) {
    l ^= p[i];
    for (int k = 0; k < 8; k++) l = (l & 1) ? (l >> 1) ^ ORACLE_POLY_REFLECTED : (l >> 1);
  }
  return l ^ 0xffffffffu;
}

static inline uint32_t load_le32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* Portable path: byte steps to a 4-byte boundary, slice-by-4 words, byte tail
 * (same shape as ExtendImpl<Slow_CRC32>, util/crc32c.cc:326-398). */
uint32_t oracle_crc32c_extend(uint32_t init, const void* data, size_t n) {
  oracle_build_tables();
  const uint8_t* p = (const uint8_t*)data;
  const uint8_t* e = p + n;
  uint32_t l = init ^ 0xffffffffu;
  while (p != e && ((uintptr_t)p & 3)) {
    l = g_tab[0][(l ^ *p++) & 0xff] ^ (l >> 8);
  }
  while ((size_t)(e - p) >= 4) {
    uint32_t c = l ^ load_le32(p);
    p += 4;
    l = g_tab[3][c & 0xff] ^ g_tab[2][(c >> 8) & 0xff] ^ g_tab[1][(c >> 16) & 0xff] ^ g_tab[0][c >> 24];
  }
  while (p != e) {
    l = g_tab[0][(l ^ *p++) & 0xff] ^ (l >> 8);
  }
  return l ^ 0xffffffffu;
}

uint32_t oracle_crc32c_value(const void* data, size_t n) { return oracle_crc32c_extend(0, data, n); }

/* util/crc32c.h:38-47 */
uint32_t oracle_crc32c_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }
uint32_t oracle_crc32c_unmask(uint32_t masked) {
  uint32_t rot = masked - 0xa282ead8u;
  return (rot >> 17) | (rot << 15);
}

/* ---- hardware path (the class of code the reference runs on x86) ---------- */

/* a(x)*b(x) mod P with the register convention bit31 = x^0. */
static uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 31; i >= 0; i--) {
    if ((a >> i) & 1) p ^= b;
    b = (b & 1) ? (b >> 1) ^ ORACLE_POLY_REFLECTED : (b >> 1);
  }
  return p;
}

/* x^e mod P */
static uint32_t gf2_xpow(uint64_t e) {
  uint32_t r = 0x80000000u; /* 1 */
  uint32_t sq = 0x40000000u; /* x */
  while (e) {
    if (e & 1) r = gf2_mulmod(r, sq);
    sq = gf2_mulmod(sq, sq);
    e >>= 1;
  }
  return r;
}

#define HW_STREAM_QWORDS 128 /* three streams of 1 KiB, like the reference's largest block */

static uint32_t g_hw_ka, g_hw_kb;
static int g_hw_ready = 0;

int oracle_crc32c_hw_available(void) {
#if ORACLE_X86
  return __builtin_cpu_supports("sse4.2") && __builtin_cpu_supports("pclmul");
#else
  return 0;
#endif
}

#if ORACLE_X86
__attribute__((target("sse4.2,pclmul")))
static uint32_t extend_hw_impl(uint32_t init, const uint8_t* p, size_t n) {
  if (!g_hw_ready) {
    /* stream A is followed by 2 streams, stream B by one: shift amounts in bits
     * minus the 33 that the multiply-and-reduce step itself contributes. */
    g_hw_ka = gf2_xpow((uint64_t)2 * HW_STREAM_QWORDS * 64 - 33);
    g_hw_kb = gf2_xpow((uint64_t)1 * HW_STREAM_QWORDS * 64 - 33);
    g_hw_ready = 1;
  }
  uint64_t l = init ^ 0xffffffffu;
  while (n && ((uintptr_t)p & 7)) {
    l = _mm_crc32_u8((uint32_t)l, *p++);
    n--;
  }
  const size_t blk = (size_t)3 * HW_STREAM_QWORDS * 8;
  const __m128i ka = _mm_cvtsi32_si128((int)g_hw_ka);
  const __m128i kb = _mm_cvtsi32_si128((int)g_hw_kb);
  while (n >= blk) {
    const uint64_t* a = (const uint64_t*)p;
    const uint64_t* b = a + HW_STREAM_QWORDS;
    const uint64_t* c = b + HW_STREAM_QWORDS;
    uint64_t ca = l, cb = 0, cc = 0;
    for (int i = 0; i < HW_STREAM_QWORDS - 1; i++) {
      ca = _mm_crc32_u64(ca, a[i]);
      cb = _mm_crc32_u64(cb, b[i]);
      cc = _mm_crc32_u64(cc, c[i]);
    }
    ca = _mm_crc32_u64(ca, a[HW_STREAM_QWORDS - 1]);
    cb = _mm_crc32_u64(cb, b[HW_STREAM_QWORDS - 1]);
    __m128i pa = _mm_clmulepi64_si128(_mm_cvtsi32_si128((int)(uint32_t)ca), ka, 0);
    __m128i pb = _mm_clmulepi64_si128(_mm_cvtsi32_si128((int)(uint32_t)cb), kb, 0);
    uint64_t fold = (uint64_t)_mm_cvtsi128_si64(_mm_xor_si128(pa, pb));
    l = _mm_crc32_u64(cc, c[HW_STREAM_QWORDS - 1] ^ fold);
    p += blk;
    n -= blk;
  }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    l = _mm_crc32_u64(l, v);
    p += 8;
    n -= 8;
  }
  while (n) {
    l = _mm_crc32_u8((uint32_t)l, *p++);
    n--;
  }
  return (uint32_t)l ^ 0xffffffffu;
}
#endif

/* Same results as oracle_crc32c_extend; SSE4.2 crc32 + PCLMULQDQ three-way. */
uint32_t oracle_crc32c_extend_hw(uint32_t init, const void* data, size_t n) {
#if ORACLE_X86
  if (oracle_crc32c_hw_available()) return extend_hw_impl(init, (const uint8_t*)data, n);
#endif
  return oracle_crc32c_extend(init, data, n);
}

/* ---- the two call shapes of the hot path ---------------------------------- */

/* table/block_based_table_builder.cc:598-601:
 *   crc = Value(block); crc = Extend(crc, &type, 1); trailer = Mask(crc) */
uint32_t oracle_sst_trailer_crc(const void* block, size_t n, uint8_t type, int use_hw) {
  uint32_t crc = use_hw ? oracle_crc32c_extend_hw(0, block, n) : oracle_crc32c_extend(0, block, n);
  crc = oracle_crc32c_extend(crc, &type, 1);
  return oracle_crc32c_mask(crc);
}

/* Batch helper for tests and for the timed CPU baseline: n_blocks blocks of
 * block_len bytes at stride, uniform type byte, masked trailers to out. */
void oracle_sst_trailer_batch(const void* base, size_t block_len, size_t stride, size_t n_blocks,
                              uint8_t type, int use_hw, uint32_t* out) {
  const uint8_t* p = (const uint8_t*)base;
  for (size_t i = 0; i < n_blocks; i++) out[i] = oracle_sst_trailer_crc(p + i * stride, block_len, type, use_hw);
}

/* table/block_fetcher.cc:41-48,64: stored = Unmask(fixed32 at data+size+1),
 * actual = Value(data, size+1); returns 1 when they match.  Both values are
 * handed back because the reference prints them in its corruption message. */
int oracle_sst_check_block(const void* data, size_t block_size, uint32_t* actual_out, uint32_t* stored_out) {
  const uint8_t* p = (const uint8_t*)data;
  uint32_t stored = oracle_crc32c_unmask(load_le32(p + block_size + 1));
  uint32_t actual = oracle_crc32c_extend(0, p, block_size + 1);
  if (actual_out) *actual_out = actual;
  if (stored_out) *stored_out = stored;
  return actual == stored;
}

/* Generic descriptor batch: out[i] = Extend(init[i], base+off[i], len[i]),
 * optionally followed by one suffix byte and Mask. */
void oracle_batch_extend(const void* base, const uint64_t* off, const uint32_t* len, const uint32_t* init,
                         const uint8_t* suffix, int masked, size_t n, uint32_t* out) {
  const uint8_t* p = (const uint8_t*)base;
  for (size_t i = 0; i < n; i++) {
    uint32_t c = oracle_crc32c_extend(init ? init[i] : 0, p + off[i], len[i]);
    if (suffix) c = oracle_crc32c_extend(c, suffix + i, 1);
    out[i] = masked ? oracle_crc32c_mask(c) : c;
  }
}

/* Deterministic fill used by tests and bench on both sides: splitmix64 stream,
 * little-endian 8-byte words. n must be a multiple of 8. */
void oracle_fill_splitmix64(void* dst, size_t n, uint64_t state) {
  uint8_t* p = (uint8_t*)dst;
  for (size_t i = 0; i + 8 <= n; i += 8) {
    state += 0x9e3779b97f4a7c15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    memcpy(p + i, &z, 8);
  }
}

/* The buffer of util/crc32c_test.cc:146-164: word 0 is zero, every later
 * 8-byte word is the FNV-style hash of the 8 bytes before it. */
void oracle_fill_fnv_chain(void* dst, size_t n) {
  uint8_t* buf = (uint8_t*)dst;
  memset(buf, 0, n < 8 ? n : 8);
  for (size_t w = 1; w * 8 + 8 <= n; w++) {
    uint64_t h = 14695981039346656037ull;
    const signed char* s = (const signed char*)(buf + (w - 1) * 8);
    for (int i = 0; i < 8; i++) {
      h += (h << 1) + (h << 4) + (h << 5) + (h << 7) + (h << 8) + (h << 40);
      h ^= (uint64_t)(int64_t)s[i];
    }
    memcpy(buf + w * 8, &h, 8);
  }
}
