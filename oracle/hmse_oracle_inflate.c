/*
 * hmse_oracle_inflate.c — CPU ORACLE for the read path.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates the decoder the reference calls through miniz: mz_inflateInit2(&s, 15) + mz_inflate(&s, MZ_FINISH)
 * (README.md:2397-2400), i.e. an RFC 1951 raw-DEFLATE decoder (README.md:2763-2765), with a preset dictionary for
 * DeltaChunk records (README.md:2182-2198).  miniz is not in the image; the decoded bytes of a valid stream are fixed
 * by RFC 1951, and the set of streams REJECTED follows stock zlib 1.2.11 (inflate.c / inftrees.c), which
 * tests/test_oracle.py pins this file against: invalid block type, stored LEN/NLEN mismatch, > 286 length or > 30
 * distance symbols, over-subscribed or incomplete code sets (an incomplete set is accepted only when its longest code
 * has length 1), a repeat code with no previous length, missing end-of-block code, an undefined code, a distance
 * beyond the window, and — the storage contract of this path — output longer or shorter than the recorded raw
 * length, or a stream that does not end in its last byte.
 */
#include <string.h>
#include "hmse_oracle.h"

typedef struct {
  const uint8_t* p;
  uint64_t len, bitpos; /* absolute bit position */
  int over;             /* read past the end */
} ibits;

static uint32_t ib_take(ibits* b, uint32_t k) {
  uint32_t v = 0;
  for (uint32_t i = 0; i < k; i++) {
    uint64_t byte = b->bitpos >> 3;
    uint32_t bit = 0;
    if (byte < b->len) bit = (b->p[byte] >> (b->bitpos & 7)) & 1u; else b->over = 1;
    v |= bit << i;
    b->bitpos++;
  }
  return v;
}

typedef struct { uint16_t count[16]; uint16_t sym[288]; } ihuff;

/* canonical table from lengths; returns 0 ok, -1 over-subscribed, -2 incomplete (and not the allowed case) */
static int ih_build(ihuff* h, const uint8_t* lens, uint32_t n, int is_codes) {
  uint16_t offs[16];
  memset(h->count, 0, sizeof h->count);
  for (uint32_t s = 0; s < n; s++) h->count[lens[s]]++;
  int max = 15;
  while (max >= 1 && h->count[max] == 0) max--;
  if (max == 0) { h->count[0] = (uint16_t)n; return 0; } /* no codes at all: legal, any use is an undefined code */
  int left = 1;
  for (int l = 1; l <= 15; l++) { left <<= 1; left -= h->count[l]; if (left < 0) return -1; }
  if (left > 0 && (is_codes || max != 1)) return -2;
  offs[1] = 0;
  for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h->count[l]);
  for (uint32_t s = 0; s < n; s++) if (lens[s]) h->sym[offs[lens[s]]++] = (uint16_t)s;
  return 0;
}

/* one symbol (bit-serial canonical decode); -1 = undefined code */
static int ih_decode(ibits* b, const ihuff* h) {
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= 15; l++) {
    code |= (int)ib_take(b, 1);
    int cnt = h->count[l];
    if (code - cnt < first) return h->sym[index + (code - first)];
    index += cnt; first += cnt; first <<= 1; code <<= 1;
  }
  return -1;
}

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* returns 0 and fills out[0..raw_len) or a negative reason code */
int orc_inflate(const uint8_t* stream, uint64_t stream_len, const uint8_t* dict, uint32_t dict_len, uint8_t* out, uint32_t raw_len) {
  ibits b = {stream, stream_len, 0, 0};
  ihuff hl, hd;
  uint8_t lens[320];
  uint32_t pos = 0;
  if (dict_len > 32768) { dict += dict_len - 32768; dict_len = 32768; }
  for (int last = 0; !last;) {
    last = (int)ib_take(&b, 1);
    uint32_t type = ib_take(&b, 2);
    if (type == 3) return -1;
    if (type == 0) {
      b.bitpos = (b.bitpos + 7) & ~7ull;
      uint32_t len = ib_take(&b, 16), nlen = ib_take(&b, 16);
      if (b.over || (len ^ nlen) != 0xFFFFu) return -2;
      uint64_t at = b.bitpos >> 3;
      if (at + len > stream_len) return -3;
      if (pos + len > raw_len) return -4;
      memcpy(out + pos, stream + at, len);
      pos += len; b.bitpos += 8ull * len;
      continue;
    }
    if (type == 1) {
      for (int s = 0; s < 288; s++) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
      ih_build(&hl, lens, 288, 0);
      for (int s = 0; s < 30; s++) lens[s] = 5;   /* zlib's fixed table has 32 distance codes, 30 and 31 invalid */
      lens[30] = lens[31] = 5;
      ih_build(&hd, lens, 32, 0);
    } else {
      uint32_t nlit = ib_take(&b, 5) + 257, ndist = ib_take(&b, 5) + 1, ncl = ib_take(&b, 4) + 4;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      if (nlit > 286 || ndist > 30) return -5;
      memset(lens, 0, 19);
      for (uint32_t i = 0; i < ncl; i++) lens[order[i]] = (uint8_t)ib_take(&b, 3);
      ihuff hc;
      if (ih_build(&hc, lens, 19, 1) != 0) return -6;
      uint32_t i = 0, tot = nlit + ndist;
      while (i < tot) {
        int s = ih_decode(&b, &hc);
        if (s < 0 || b.over) return -7;
        uint32_t rep = 1, val = (uint32_t)s;
        if (s == 16) { if (i == 0) return -8; val = lens[i - 1]; rep = 3 + ib_take(&b, 2); }
        else if (s == 17) { val = 0; rep = 3 + ib_take(&b, 3); }
        else if (s == 18) { val = 0; rep = 11 + ib_take(&b, 7); }
        if (i + rep > tot) return -9;
        while (rep--) lens[i++] = (uint8_t)val;
      }
      if (lens[256] == 0) return -10;
      if (ih_build(&hl, lens, nlit, 0) != 0) return -11;
      if (ih_build(&hd, lens + nlit, ndist, 0) != 0) return -12;
    }
    for (;;) {
      int s = ih_decode(&b, &hl);
      if (s < 0 || b.over) return -13;
      if (s < 256) { if (pos >= raw_len) return -4; out[pos++] = (uint8_t)s; continue; }
      if (s == 256) break;
      if (s > 285) return -14;
      uint32_t len = LBASE[s - 257] + ib_take(&b, LEXT[s - 257]);
      int ds = ih_decode(&b, &hd);
      if (ds < 0 || ds > 29 || b.over) return -15;
      uint32_t dist = DBASE[ds] + ib_take(&b, DEXT[ds]);
      if (b.over) return -13;
      if (dist > pos + dict_len) return -16;
      if (pos + len > raw_len) return -4;
      for (uint32_t i = 0; i < len; i++, pos++) {
        int64_t sp = (int64_t)pos - dist;
        out[pos] = sp >= 0 ? out[sp] : dict[(int64_t)dict_len + sp];
      }
    }
  }
  if (b.over) return -13;
  if (pos != raw_len) return -17;
  if (((b.bitpos + 7) >> 3) != stream_len) return -18;
  return 0;
}

/* batch form mirroring hmse_l1_inflate: returns the number of corrupt streams; ok[k] = 1 where chunk k decoded */
uint64_t orc_inflate_chunks(const uint8_t* streams, const uint64_t* stream_off, const uint32_t* stream_len, const uint8_t* kind,
                            const int64_t* base, uint64_t n_sel, const uint64_t* raw_off, uint8_t* raw_out, uint8_t* ok) {
  uint64_t bad = 0;
  for (uint64_t k = 0; k < n_sel; k++) {
    const uint64_t s0 = stream_off[k], sl = stream_len ? stream_len[k] : stream_off[k + 1] - s0;
    const uint8_t* dict = 0; uint32_t dl = 0; int r = 0;
    if (kind[k] == HMSE_KIND_DELTA) {
      const int64_t bb = base ? base[k] : -1;
      if (bb < 0 || (uint64_t)bb >= k || (ok && !ok[bb])) r = -20;
      else { dict = raw_out + raw_off[bb]; dl = (uint32_t)(raw_off[bb + 1] - raw_off[bb]); }
    }
    if (r == 0) r = orc_inflate(streams + s0, sl, dict, dl, raw_out + raw_off[k], (uint32_t)(raw_off[k + 1] - raw_off[k]));
    if (ok) ok[k] = r == 0;
    bad += r != 0;
  }
  return bad;
}
