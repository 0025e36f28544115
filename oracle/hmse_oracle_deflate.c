/*
 * hmse_oracle_deflate.c — CPU ORACLE (test infrastructure) for L1: per-chunk raw DEFLATE
 * (RFC 1951) with a preset dictionary.
 *
 * The reference's codec is miniz level 9, window 15 (README.md:2374, 2378) and its delta step
 * is resolved by SURVEY.md D6 into "DEFLATE with zdict = base chunk".  zlib/miniz's lazy,
 * position-serial matcher cannot be reproduced bit-for-bit by a parallel encoder, so this file
 * DEFINES the build's deterministic encoder (SURVEY.md §7 route A); the HIP kernel must emit
 * byte-identical streams, every stream must inflate through stock zlib, and zlib-9's size is
 * reported next to it.
 *
 * Encoder definition (W = dict ++ chunk, T = |W| <= 65536, chunk starts at Dl):
 *  1. h(q) = (LE32(W+q) * 0x9E3779B1) >> 20 for q+4 <= T; bucket lists hold positions ascending.
 *  2. For every chunk position p (p+4 <= T): walk the bucket backwards from p's predecessor,
 *     at most D candidates, stop at distance > 32768; ml = common prefix capped at
 *     min(258, T-p); keep the longest (first found wins ties); accept iff >= 4.
 *  2b. Dictionary jobs only (Dl > 0) — diagonal anchors.  A chunk with an LSH base is a near-duplicate
 *     of it, so most positions have a full-length match in the dictionary on the same diagonal as
 *     their neighbours.  Every 64th chunk position pa = Dl + 64a (pa+4 <= T) is an ANCHOR: among
 *     the (at most) 64 nearest dictionary members of its bucket within distance 32768, the one
 *     agreeing with pa on most of the next 32 bytes (window padded with zeros; ties: nearest), if
 *     all 32 agree, gives the anchor's distance d, and run = number of x < min(512, T-pa) with
 *     W[pa+x] == W[pa+x-d] for all smaller x too (the run along the diagonal).  A position p in
 *     block a takes a HINT from anchor a, else a-1: q' = p-d must lie in the dictionary, its known
 *     common prefix with p is m = min(258, T-p) if run = 512, else min(pa+run-p, 258, T-p) if
 *     pa+run > p; the longer hint wins (anchor a on ties).
 *  2c. (round 3; replaces the take rule of 2b) If m >= 16, position p TAKES the hinted match (m, p-q') outright: no walk
 *     at all.  85 % of the positions of a dictionary job are of this kind (70 % with the full length min(258, T-p), the
 *     rest next to an edit); their walks over the chunk's own candidates only ever served the FULL encoding of the
 *     same chunk, which rule 7 no longer needs in the common case.  Positions without a hint: rule 2 unchanged
 *     (chunk and dictionary candidates, nearest first, depth D).  Measured against walking everything (609 dictionary
 *     jobs of three corpus profiles): DELTA bytes -0.5 % — the hinted matches share one distance, which codes cheaply.
 *  3. Parse from p = Dl: take the match at p iff mlen[p] >= 4 and not (mlen[p+1] > mlen[p]);
 *     otherwise emit the literal and move to p+1 (one-step lazy evaluation, like deflate_slow).
 *  4. One final block: stored / fixed / dynamic, whichever is smallest in bits
 *     (ties: stored, then fixed).
 *  5. Huffman lengths: symbols sorted by (freq, index); two-queue Huffman (leaf wins ties);
 *     the multiset of depths is clamped to the limit with the Kraft repair below and handed out
 *     in sorted order (most frequent = shortest); fewer than two used symbols -> the lowest
 *     unused indices get frequency 1.  Canonical codes per RFC 1951 §3.2.2.
 *  6. Code-length RLE per tree (lit/len, then dist; runs do not cross): zero runs -> 18 (11..138)
 *     while >= 11, then 17 (3..10), else literal zeros; non-zero runs -> the value once, then 16
 *     (3..6) while >= 3 remain, then literals.
 *  7. FULL or DELTA (orc_deflate_chunks; README.md:1328, 2175 "delta stored only if <= 20 % of the original chunk",
 *     SURVEY.md D7 "net savings"): a chunk with a base is encoded against it first (n2 bytes).  If
 *     5*n2 <= chunk length — the reference's own criterion — the record is DELTA and the chunk's FULL
 *     encoding is never computed; a larger delta is kept iff it nets savings over FULL after the 8-byte
 *     DeltaChunk header (n2 + 8 < n1).  With cfg->delta_max_ratio_pct != 0 a delta above that share of
 *     the chunk is refused outright.
 */
#include "hmse_oracle.h"
#include <stdlib.h>
#include <string.h>

#define HB 12
#define NBUCKET (1u << HB)
#define MINM 4u
#define MAXM 258u
#define WMAX 32768u

static uint32_t le32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint32_t hash4(uint32_t x) { return (x * 0x9E3779B1u) >> (32 - HB); }

static uint32_t depth_from_cfg(const hmse_cfg* cfg) {
  static const uint32_t tab[10] = {2, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  if (cfg->chain_depth) return cfg->chain_depth;
  uint32_t lv = cfg->level > 9 ? 9 : cfg->level;
  return tab[lv];
}

uint32_t orc_deflate_bound(uint32_t len) { return len + 5; }

/* ---- matching ---------------------------------------------------------------------------- */

typedef struct {
  uint32_t T, Dl;
  uint8_t* W;
  uint16_t* mlen;  /* per chunk position */
  uint16_t* mdist;
} lz_t;

static void lz_build(lz_t* z, const uint8_t* chunk, uint32_t len, const uint8_t* dict, uint32_t dlen,
                     const hmse_cfg* cfg) {
  if (dlen > WMAX) { dict += dlen - WMAX; dlen = WMAX; }
  uint32_t T = dlen + len;
  z->T = T; z->Dl = dlen;
  z->W = (uint8_t*)malloc(T + 64);
  if (dlen) memcpy(z->W, dict, dlen);
  if (len) memcpy(z->W + dlen, chunk, len);
  memset(z->W + T, 0, 64);
  z->mlen = (uint16_t*)calloc(len + 1, 2);
  z->mdist = (uint16_t*)calloc(len + 1, 2);
  uint32_t nh = T >= 4 ? T - 3 : 0; /* positions with a 4-byte hash */
  uint32_t* start = (uint32_t*)calloc(NBUCKET + 1, 4);
  uint32_t* fill = (uint32_t*)calloc(NBUCKET, 4);
  uint16_t* S = (uint16_t*)malloc((nh + 1) * 2);
  uint32_t* rank = (uint32_t*)malloc((nh + 1) * 4);
  for (uint32_t q = 0; q < nh; q++) start[hash4(le32(z->W + q)) + 1]++;
  for (uint32_t h = 0; h < NBUCKET; h++) start[h + 1] += start[h];
  for (uint32_t q = 0; q < nh; q++) { /* ascending q => ascending inside each bucket */
    uint32_t h = hash4(le32(z->W + q));
    uint32_t r = start[h] + fill[h]++;
    S[r] = (uint16_t)q; rank[q] = r;
  }
  uint32_t D = depth_from_cfg(cfg);
  /* rule 2b: anchors of a dictionary job */
  uint32_t n_anch = dlen ? (len + 63) / 64 : 0;
  uint32_t* anch_d = (uint32_t*)calloc(n_anch + 1, 4);
  uint32_t* anch_run = (uint32_t*)calloc(n_anch + 1, 4);
  for (uint32_t a = 0; a < n_anch; a++) {
    uint32_t pa = dlen + 64 * a;
    if (pa + 4 > T) continue;
    uint32_t h = hash4(le32(z->W + pa));
    uint32_t lo = start[h], hi = start[h + 1];
    uint32_t l = lo;                                   /* dictionary members are the first ones of the bucket */
    while (l < hi && S[l] < dlen) l++;
    uint32_t first = l > lo + 64 ? l - 64 : lo;
    uint32_t bml = 0, bq = 0;
    for (uint32_t idx = first; idx < l; idx++) {         /* ascending position: a later one is nearer and wins ties */
      uint32_t q = S[idx];
      if (pa - q > WMAX) continue;
      uint32_t ml = 0;
      while (ml < 32 && z->W[q + ml] == z->W[pa + ml]) ml++;   /* (W is zero-padded behind T) */
      if (ml >= bml) { bml = ml; bq = q; }
    }
    if (bml >= 32) {
      uint32_t d = pa - bq, lim = T - pa < 512 ? T - pa : 512, run = 0;
      while (run < lim && z->W[pa + run] == z->W[pa + run - d]) run++;
      anch_d[a] = d; anch_run[a] = run;
    }
  }
  for (uint32_t p = dlen; p < T; p++) {
    if (p + 4 > T) continue;
    uint32_t h = hash4(le32(z->W + p));
    uint32_t r = rank[p], g = start[h];
    uint32_t maxlen = T - p < MAXM ? T - p : MAXM;
    uint32_t best = MINM - 1, bdist = 0;
    /* hint of rule 2b */
    uint32_t hm = 0, hq = 0;
    if (dlen) {
      uint32_t a0 = (p - dlen) >> 6;
      for (uint32_t back = 0; back < 2; back++) {
        if (a0 < back) continue;
        uint32_t d = anch_d[a0 - back], run = anch_run[a0 - back], pa = dlen + ((a0 - back) << 6);
        if (d == 0 || d > p || p - d >= dlen) continue;
        uint32_t m = 0;
        if (run >= 512) m = maxlen;
        else if (pa + run > p) m = pa + run - p < maxlen ? pa + run - p : maxlen;
        if (m > hm) { hm = m; hq = p - d; }
      }
    }
    if (hm >= 16) {                                       /* rule 2c: a hinted match is taken outright */
      best = hm; bdist = p - hq;
    } else {
      for (uint32_t k = 1; k <= D && r >= g + k; k++) {
        uint32_t q = S[r - k];
        if (p - q > WMAX) break;
        uint32_t ml = 0;
        while (ml < maxlen && z->W[q + ml] == z->W[p + ml]) ml++;
        if (ml > best) { best = ml; bdist = p - q; if (ml == maxlen) break; }
      }
    }
    if (best >= MINM) { z->mlen[p - dlen] = (uint16_t)best; z->mdist[p - dlen] = (uint16_t)bdist; }
  }
  free(anch_d); free(anch_run);
  free(start); free(fill); free(S); free(rank);
}

static void lz_free(lz_t* z) { free(z->W); free(z->mlen); free(z->mdist); }

void orc_deflate_matches(const uint8_t* chunk, uint32_t len, const uint8_t* dict, uint32_t dlen,
                         const hmse_cfg* cfg, uint16_t* mlen, uint16_t* mdist) {
  lz_t z; lz_build(&z, chunk, len, dict, dlen, cfg);
  memcpy(mlen, z.mlen, (size_t)len * 2); memcpy(mdist, z.mdist, (size_t)len * 2);
  lz_free(&z);
}

/* ---- symbol maps (RFC 1951 §3.2.5, closed forms) ----------------------------------------- */

static int ilog2(uint32_t v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }

static void len_sym(uint32_t len, uint32_t* code, uint32_t* ebits, uint32_t* eval) {
  if (len == 258) { *code = 285; *ebits = 0; *eval = 0; return; }
  uint32_t l = len - 3;
  if (l < 8) { *code = 257 + l; *ebits = 0; *eval = 0; return; }
  uint32_t e = (uint32_t)ilog2(l) - 2;
  *code = 257 + 4 * e + 4 + ((l >> e) & 3); *ebits = e; *eval = l & ((1u << e) - 1);
}
static void dist_sym(uint32_t dist, uint32_t* code, uint32_t* ebits, uint32_t* eval) {
  uint32_t d = dist - 1;
  if (d < 4) { *code = d; *ebits = 0; *eval = 0; return; }
  uint32_t hb = (uint32_t)ilog2(d), e = hb - 1;
  *code = 2 * hb + ((d >> e) & 1); *ebits = e; *eval = d & ((1u << e) - 1);
}

/* ---- Huffman ------------------------------------------------------------------------------- */

typedef struct { uint32_t f; uint32_t s; } fs_t;
static int fs_cmp(const void* a, const void* b) {
  const fs_t* x = (const fs_t*)a; const fs_t* y = (const fs_t*)b;
  if (x->f != y->f) return x->f < y->f ? -1 : 1;
  return (x->s > y->s) - (x->s < y->s);
}

static void huff_lengths(const uint32_t* freq_in, uint32_t n, uint32_t limit, uint8_t* lens) {
  uint32_t freq[288];
  memcpy(freq, freq_in, n * 4);
  memset(lens, 0, n);
  uint32_t used = 0;
  for (uint32_t i = 0; i < n; i++) used += freq[i] != 0;
  for (uint32_t i = 0; used < 2 && i < n; i++) if (!freq[i]) { freq[i] = 1; used++; }
  fs_t leaf[288]; uint32_t m = 0;
  for (uint32_t i = 0; i < n; i++) if (freq[i]) { leaf[m].f = freq[i]; leaf[m].s = i; m++; }
  qsort(leaf, m, sizeof *leaf, fs_cmp);
  /* two-queue merge; node ids: leaves 0..m-1, internals m..2m-2 */
  uint64_t w[576]; uint32_t parent[576];
  for (uint32_t i = 0; i < m; i++) w[i] = leaf[i].f;
  uint32_t li = 0, ii = m, ni = m;
  for (uint32_t step = 0; step + 1 < m; step++) {
    uint32_t pick[2];
    for (int k = 0; k < 2; k++) {
      int take_leaf;
      if (li < m && ii < ni) take_leaf = w[li] <= w[ii]; /* leaf wins ties */
      else take_leaf = li < m;
      pick[k] = take_leaf ? li++ : ii++;
    }
    w[ni] = w[pick[0]] + w[pick[1]];
    parent[pick[0]] = ni; parent[pick[1]] = ni; ni++;
  }
  uint32_t root = ni - 1;
  uint32_t depth[576];
  depth[root] = 0;
  for (uint32_t v = root; v-- > 0;) depth[v] = depth[parent[v]] + 1; /* parents have larger ids */
  uint32_t cnt[64]; memset(cnt, 0, sizeof cnt);
  for (uint32_t i = 0; i < m; i++) { uint32_t d = depth[i]; if (d > limit) d = limit; cnt[d]++; }
  /* Kraft repair (miniz-style enforce-max-code-size) */
  uint32_t total = 0;
  for (uint32_t i = limit; i > 0; i--) total += cnt[i] << (limit - i);
  while (total > (1u << limit)) {
    cnt[limit]--;
    for (uint32_t i = limit - 1; i > 0; i--) if (cnt[i]) { cnt[i]--; cnt[i + 1] += 2; break; }
    total--;
  }
  /* most frequent symbols (end of the sorted list) get the shortest lengths */
  uint32_t j = m;
  for (uint32_t l = 1; l <= limit; l++) for (uint32_t c = cnt[l]; c > 0; c--) lens[leaf[--j].s] = (uint8_t)l;
}

static uint32_t bitrev(uint32_t v, uint32_t n) { uint32_t r = 0; for (uint32_t i = 0; i < n; i++) { r = (r << 1) | (v & 1); v >>= 1; } return r; }

static void huff_codes(const uint8_t* lens, uint32_t n, uint16_t* codes) {
  uint32_t bl[16] = {0}, next[16] = {0};
  for (uint32_t i = 0; i < n; i++) bl[lens[i]]++;
  bl[0] = 0;
  uint32_t code = 0;
  for (uint32_t b = 1; b <= 15; b++) { code = (code + bl[b - 1]) << 1; next[b] = code; }
  for (uint32_t i = 0; i < n; i++) codes[i] = lens[i] ? (uint16_t)bitrev(next[lens[i]]++, lens[i]) : 0;
}

/* ---- bit writer ------------------------------------------------------------------------------ */

typedef struct { uint8_t* out; uint64_t cap, pos; uint64_t acc; uint32_t nb; int ovf; } bw_t;
static void bw_put(bw_t* b, uint32_t v, uint32_t n) {
  b->acc |= (uint64_t)v << b->nb; b->nb += n;
  while (b->nb >= 8) { if (b->pos < b->cap) b->out[b->pos] = (uint8_t)b->acc; else b->ovf = 1; b->pos++; b->acc >>= 8; b->nb -= 8; }
}
static void bw_flush(bw_t* b) { if (b->nb) bw_put(b, 0, 8 - b->nb); }

/* ---- code-length RLE ------------------------------------------------------------------------- */

typedef struct { uint8_t sym; uint8_t ebits; uint8_t eval; } clt_t;
static uint32_t rle_lengths(const uint8_t* l, uint32_t n, clt_t* out) {
  uint32_t k = 0, i = 0;
  while (i < n) {
    uint32_t j = i + 1;
    while (j < n && l[j] == l[i]) j++;
    uint32_t run = j - i, v = l[i];
    if (v == 0) {
      while (run >= 11) { uint32_t c = run > 138 ? 138 : run; out[k].sym = 18; out[k].ebits = 7; out[k].eval = (uint8_t)(c - 11); k++; run -= c; }
      if (run >= 3) { out[k].sym = 17; out[k].ebits = 3; out[k].eval = (uint8_t)(run - 3); k++; run = 0; }
      while (run--) { out[k].sym = 0; out[k].ebits = 0; out[k].eval = 0; k++; }
    } else {
      out[k].sym = (uint8_t)v; out[k].ebits = 0; out[k].eval = 0; k++; run--;
      while (run >= 3) { uint32_t c = run > 6 ? 6 : run; out[k].sym = 16; out[k].ebits = 2; out[k].eval = (uint8_t)(c - 3); k++; run -= c; }
      while (run--) { out[k].sym = (uint8_t)v; out[k].ebits = 0; out[k].eval = 0; k++; }
    }
    i = j;
  }
  return k;
}

static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* ---- encoder --------------------------------------------------------------------------------- */

/* rules 3-6: parse the per-position matches, choose the block type, emit the stream */
static int64_t encode_matches(const uint8_t* chunk, uint32_t len, const uint16_t* zmlen, const uint16_t* zmdist,
                              uint8_t* out, uint64_t out_cap) {
  /* parse */
  uint16_t* tl = (uint16_t*)malloc(((size_t)len + 1) * 2); /* literal byte or match length */
  uint16_t* td = (uint16_t*)malloc(((size_t)len + 1) * 2); /* 0 = literal, else distance (65536 never occurs: <= 32768) */
  uint32_t nt = 0;
  for (uint32_t p = 0; p < len;) {
    uint32_t ml = zmlen[p];
    if (ml >= MINM && !(p + 1 < len && zmlen[p + 1] > ml)) { tl[nt] = (uint16_t)ml; td[nt] = zmdist[p]; nt++; p += ml; }
    else { tl[nt] = chunk[p]; td[nt] = 0; nt++; p++; }
  }
  /* frequencies */
  uint32_t lf[288], df[32];
  memset(lf, 0, sizeof lf); memset(df, 0, sizeof df);
  uint64_t extra_bits = 0;
  for (uint32_t i = 0; i < nt; i++) {
    if (td[i]) { uint32_t c, eb, ev; len_sym(tl[i], &c, &eb, &ev); lf[c]++; extra_bits += eb; dist_sym(td[i], &c, &eb, &ev); df[c]++; extra_bits += eb; }
    else lf[tl[i]]++;
  }
  lf[256]++;
  /* fixed cost */
  uint64_t fixed_bits = 3 + extra_bits;
  for (uint32_t s = 0; s < 286; s++) { uint32_t l = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8; fixed_bits += (uint64_t)lf[s] * l; }
  for (uint32_t s = 0; s < 30; s++) fixed_bits += (uint64_t)df[s] * 5;
  /* dynamic trees */
  uint8_t ll[288], dl[32], cl[19]; uint16_t lc[288], dc[32], cc[19];
  huff_lengths(lf, 286, 15, ll); huff_lengths(df, 30, 15, dl);
  uint32_t nlit = 286; while (nlit > 257 && ll[nlit - 1] == 0) nlit--;
  uint32_t ndist = 30; while (ndist > 1 && dl[ndist - 1] == 0) ndist--;
  clt_t rle[320]; uint32_t nr = rle_lengths(ll, nlit, rle);
  nr += rle_lengths(dl, ndist, rle + nr);
  uint32_t cf[19]; memset(cf, 0, sizeof cf);
  for (uint32_t i = 0; i < nr; i++) cf[rle[i].sym]++;
  huff_lengths(cf, 19, 7, cl);
  uint32_t ncl = 19; while (ncl > 4 && cl[CL_ORDER[ncl - 1]] == 0) ncl--;
  uint64_t dyn_bits = 3 + 14 + 3 * ncl + extra_bits;
  for (uint32_t i = 0; i < nr; i++) dyn_bits += cl[rle[i].sym] + rle[i].ebits;
  for (uint32_t s = 0; s < 286; s++) dyn_bits += (uint64_t)lf[s] * ll[s];
  for (uint32_t s = 0; s < 30; s++) dyn_bits += (uint64_t)df[s] * dl[s];
  uint64_t stored_bits = 8ull * (5 + (uint64_t)len);

  bw_t b; memset(&b, 0, sizeof b); b.out = out; b.cap = out_cap;
  if (stored_bits <= fixed_bits && stored_bits <= dyn_bits) {
    bw_put(&b, 1, 1); bw_put(&b, 0, 2); bw_flush(&b);
    bw_put(&b, len & 0xFFFF, 16); bw_put(&b, (~len) & 0xFFFF, 16);
    for (uint32_t i = 0; i < len; i++) bw_put(&b, chunk[i], 8);
  } else {
    int fixed = fixed_bits <= dyn_bits;
    if (fixed) {
      for (uint32_t s = 0; s < 288; s++) ll[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
      for (uint32_t s = 0; s < 32; s++) dl[s] = 5;
      huff_codes(ll, 288, lc); huff_codes(dl, 32, dc);
      bw_put(&b, 1, 1); bw_put(&b, 1, 2);
    } else {
      huff_codes(ll, 286, lc); huff_codes(dl, 30, dc); huff_codes(cl, 19, cc);
      bw_put(&b, 1, 1); bw_put(&b, 2, 2);
      bw_put(&b, nlit - 257, 5); bw_put(&b, ndist - 1, 5); bw_put(&b, ncl - 4, 4);
      for (uint32_t i = 0; i < ncl; i++) bw_put(&b, cl[CL_ORDER[i]], 3);
      for (uint32_t i = 0; i < nr; i++) { bw_put(&b, cc[rle[i].sym], cl[rle[i].sym]); if (rle[i].ebits) bw_put(&b, rle[i].eval, rle[i].ebits); }
    }
    for (uint32_t i = 0; i < nt; i++) {
      if (td[i]) {
        uint32_t c, eb, ev;
        len_sym(tl[i], &c, &eb, &ev); bw_put(&b, lc[c], ll[c]); if (eb) bw_put(&b, ev, eb);
        dist_sym(td[i], &c, &eb, &ev); bw_put(&b, dc[c], dl[c]); if (eb) bw_put(&b, ev, eb);
      } else bw_put(&b, lc[tl[i]], ll[tl[i]]);
    }
    bw_put(&b, lc[256], ll[256]);
    bw_flush(&b);
  }
  free(tl); free(td);
  if (b.ovf) return HMSE_ENOSPC;
  return (int64_t)b.pos;
}

int64_t orc_deflate(const uint8_t* chunk, uint32_t len, const uint8_t* dict, uint32_t dlen,
                    const hmse_cfg* cfg, uint8_t* out, uint64_t out_cap) {
  if (len > 65535u) return HMSE_EINVAL;
  lz_t z; lz_build(&z, chunk, len, dict, dlen, cfg);
  int64_t n = encode_matches(chunk, len, z.mlen, z.mdist, out, out_cap);
  lz_free(&z);
  return n;
}

/* Batch form (mirrors hmse_l1_deflate), rule 7: with a base the dictionary stream comes first; it is the record
 * (DELTA) when it is at most a fifth of the chunk (README.md:1328, 2175); otherwise FULL is computed and DELTA kept
 * iff it nets savings after the 8-byte DeltaChunk header (README.md:2182-2189); with
 * cfg->delta_max_ratio_pct != 0, never if delta_len*100 > pct*chunk_len. */
int orc_deflate_chunks(const uint8_t* data, const uint64_t* cuts, const uint64_t* chunk_ids,
                       const int64_t* base, uint64_t n_sel, const hmse_cfg* cfg, uint8_t* out,
                       uint64_t out_cap, uint64_t* out_off, uint8_t* kind) {
  uint64_t pos = 0; int ovf = 0;
  uint8_t* tmp1 = (uint8_t*)malloc(65536 + 16); uint8_t* tmp2 = (uint8_t*)malloc(65536 + 16);
  for (uint64_t k = 0; k < n_sel; k++) {
    uint64_t c = chunk_ids ? chunk_ids[k] : k;
    const uint8_t* p = data + cuts[c]; uint32_t len = (uint32_t)(cuts[c + 1] - cuts[c]);
    const uint8_t* best = NULL; int64_t bn = 0; uint8_t kd = HMSE_KIND_FULL;
    if (base && base[k] >= 0) {
      uint64_t bc = chunk_ids ? chunk_ids[base[k]] : (uint64_t)base[k];
      const uint8_t* d = data + cuts[bc]; uint32_t dl = (uint32_t)(cuts[bc + 1] - cuts[bc]);
      int64_t n2 = orc_deflate(p, len, d, dl, cfg, tmp2, 65536 + 16);
      int gate = !(cfg->delta_max_ratio_pct && (uint64_t)n2 * 100 > (uint64_t)cfg->delta_max_ratio_pct * len);
      if (n2 >= 0 && gate && (uint64_t)n2 * 5 <= len) { best = tmp2; bn = n2; kd = HMSE_KIND_DELTA; }   /* FULL never computed */
      else {
        int64_t n1 = orc_deflate(p, len, NULL, 0, cfg, tmp1, 65536 + 16);
        if (n2 >= 0 && gate && n2 + 8 < n1) { best = tmp2; bn = n2; kd = HMSE_KIND_DELTA; }
        else { best = tmp1; bn = n1; }
      }
    } else {
      best = tmp1; bn = orc_deflate(p, len, NULL, 0, cfg, tmp1, 65536 + 16);
    }
    if (bn < 0) { free(tmp1); free(tmp2); return (int)bn; }
    out_off[k] = pos;
    if (pos + (uint64_t)bn <= out_cap) memcpy(out + pos, best, (size_t)bn); else ovf = 1;
    pos += (uint64_t)bn;
    if (kind) kind[k] = kd;
  }
  out_off[n_sel] = pos;
  free(tmp1); free(tmp2);
  return ovf ? HMSE_ENOSPC : HMSE_OK;
}
