// GF(2) polynomial arithmetic modulo the CRC-32C generator, host side.
//
// Register convention (the one the reference's byte step implies,
// util/crc32c.cc:366-369): a 32-bit register holds a polynomial of degree < 32
// with bit 31 = x^0 and bit 0 = x^31; feeding one zero bit multiplies by x.
// Feeding n zero bytes into a register r therefore gives r * x^(8n) mod P.
#pragma once
#include <stdint.h>

namespace tcrc {

constexpr uint32_t kPolyReflected = 0x82f63b78u;  // util/crc32c.cc:83 (table0_[0x80])

// a * b mod P
inline uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 31; i >= 0; --i) {
    if ((a >> i) & 1u) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ kPolyReflected : (b >> 1);
  }
  return p;
}

// x^e mod P
inline uint32_t gf2_xpow(uint64_t e) {
  uint32_t r = 0x80000000u;   // 1
  uint32_t sq = 0x40000000u;  // x
  while (e) {
    if (e & 1u) r = gf2_mulmod(r, sq);
    sq = gf2_mulmod(sq, sq);
    e >>= 1;
  }
  return r;
}

// x^(-8n) mod P: the register that, advanced through n zero bytes, becomes 1.  x is a unit
// because P has a constant term, and dividing by x undoes one multiply-by-x step exactly.
inline uint32_t gf2_xinv_bytes(uint32_t n) {
  uint32_t r = 0x80000000u;  // 1
  for (uint32_t i = 0; i < 8 * n; ++i) {
    const uint32_t carried = r >> 31;  // the bit the forward step moved in from x^32
    if (carried) r ^= kPolyReflected;
    r = (r << 1) | carried;
  }
  return r;
}

// Register r after nbytes further zero bytes.
inline uint32_t gf2_shift_bytes(uint32_t r, uint64_t nbytes) { return gf2_mulmod(r, gf2_xpow(8 * nbytes)); }

// Lookup set for "advance a register through n zero bytes" (n >= 3):
//   Z_n(c) = t[0][c & 0xff] ^ t[1][(c >> 8) & 0xff] ^ t[2][(c >> 16) & 0xff] ^ t[3][c >> 24]
// i.e. t[k][b] = (b placed in register byte k) * x^(8n).  With n = 4 this is the
// slice-by-4 set of util/crc32c.cc:326-332 (t[3] is the one-byte table table0_).
inline void build_shift_tables(uint32_t n, uint32_t t[4][256]) {
  for (int k = 0; k < 4; ++k) {
    const uint32_t m = gf2_xpow(8ull * (n - k));
    for (uint32_t b = 0; b < 256; ++b) t[k][b] = gf2_mulmod(b, m);
  }
}

}  // namespace tcrc
