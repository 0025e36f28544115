/*
 * hmse_oracle.c — CPU ORACLE (test infrastructure, see hmse_oracle.h) for L2, L3, L4.
 * Plain C, serial, written for obviousness.  Each function cites the reference lines
 * (relative to /root/reference) whose behaviour it restates.
 */
#include "hmse_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ config */

void orc_cfg_default(hmse_cfg* c) {
  memset(c, 0, sizeof *c);
  c->struct_size = (uint32_t)sizeof *c;
  c->min_size = 2048;  /* README.md:2444-2446 ratios min=avg/4, max=4*avg at avg 8 KiB */
  c->avg_size = 8192;
  c->max_size = 32768;
  c->norm_level = 2;
  c->seg_size = 4u << 20;
  c->n_hashes = 128; /* README.md:2575 */
  c->shingle = 4;    /* README.md:2584-2586 */
  c->seed_base = 0;  /* README.md:2589 (h = 0..127), SURVEY D5 */
  c->bands = 4;      /* README.md:1987-1996 */
  c->rows = 32;
  c->band_bits = 16;
  c->level = 9;      /* README.md:2374 */
  c->chain_depth = 0;
  c->layers = HMSE_LAYER_L1 | HMSE_LAYER_L2 | HMSE_LAYER_L3 | HMSE_LAYER_L4;
  c->delta_max_ratio_pct = 0;
}

/* ------------------------------------------------------------------ L2 */

/* Gear table: splitmix64 stream from the fixed seed "HMSE_L2G". SURVEY D2: the reference
 * names Rabin/FastCDC (README.md:289) but fixes no table; this one is the build's own. */
void orc_gear_table(uint64_t t[256]) {
  uint64_t x = 0x484D53455F4C3247ull; /* "HMSE_L2G" */
  for (int i = 0; i < 256; i++) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    t[i] = z ^ (z >> 31);
  }
}

static int ilog2_u32(uint32_t v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }

/* two-mask normalisation: the hard mask (size < avg) has log2(avg)+norm TOP bits, the easy one
 * log2(avg)-norm TOP bits; easy mask bits are a subset of the hard mask's. */
void orc_cdc_masks(const hmse_cfg* cfg, uint64_t* mask_s, uint64_t* mask_l) {
  int bits = ilog2_u32(cfg->avg_size);
  int bs = bits + (int)cfg->norm_level, bl = bits - (int)cfg->norm_level;
  if (bl < 1) bl = 1;
  if (bs > 48) bs = 48;
  *mask_s = ~0ull << (64 - bs);
  *mask_l = ~0ull << (64 - bl);
}

/* README.md:2475-2490 control flow: state initialised once per segment and never reset at a cut;
 * cut after byte i iff size >= MIN && (hash hit || size >= MAX); forced cut at the segment end. */
uint64_t orc_cdc(const uint8_t* data, uint64_t n, const uint64_t* seg_off, uint32_t n_seg,
                 const hmse_cfg* cfg, uint64_t* cuts, uint64_t cuts_cap) {
  uint64_t G[256], ms, ml;
  orc_gear_table(G);
  orc_cdc_masks(cfg, &ms, &ml);
  uint64_t nc = 0;
  if (cuts && cuts_cap > 0) cuts[0] = 0;
  const uint64_t MIN = cfg->min_size, AVG = cfg->avg_size, MAX = cfg->max_size;
  for (uint32_t s = 0; s < n_seg; s++) {
    uint64_t a = seg_off[s], b = seg_off[s + 1];
    if (b > n) b = n;
    uint64_t h = 0, start = a;
    for (uint64_t i = a; i < b; i++) {
      h = (h << 1) + G[data[i]];
      uint64_t size = i + 1 - start;
      if (size < MIN) continue;
      int hit = (size < AVG) ? ((h & ms) == 0) : ((h & ml) == 0);
      if (hit || size >= MAX) {
        nc++;
        if (cuts && nc < cuts_cap) cuts[nc] = i + 1;
        start = i + 1;
      }
    }
    if (start < b) {
      nc++;
      if (cuts && nc < cuts_cap) cuts[nc] = b;
    }
  }
  return nc;
}

/* The literal skeleton: rabin_slide (README.md:2456-2464) + loop (README.md:2475-2490),
 * MIN 1024 / MAX 16384 / mask 4095 (README.md:2444-2447). Used only for the acceptance-band test. */
uint64_t orc_cdc_reference_skeleton(const uint8_t* data, uint64_t n, uint64_t* cuts, uint64_t cuts_cap) {
  uint64_t hash = 0; uint32_t window[64]; int wpos = 0;
  memset(window, 0, sizeof window);
  uint64_t nc = 0, start = 0;
  if (cuts && cuts_cap) cuts[0] = 0;
  for (uint64_t pos = 0; pos < n;) {
    uint8_t in = data[pos];
    uint8_t out = (uint8_t)window[wpos];
    window[wpos] = in; wpos = (wpos + 1) % 64;
    hash = (hash << 1) ^ in ^ ((uint64_t)out << 7);
    pos++;
    uint64_t size = pos - start;
    if (size >= 1024 && ((hash & 4095) == 0 || size >= 16384)) {
      nc++; if (cuts && nc < cuts_cap) cuts[nc] = pos;
      start = pos;
    }
  }
  if (start < n) { nc++; if (cuts && nc < cuts_cap) cuts[nc] = n; }
  return nc;
}

/* ------------------------------------------------------------------ L3 SHA-256 (FIPS 180-4) */

static const uint32_t K256[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
  0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
  0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
  0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
  0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
  0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
  0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
  0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};

static uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

static void sha256_block(uint32_t st[8], const uint8_t* p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++)
    w[i] = ((uint32_t)p[4*i] << 24) | ((uint32_t)p[4*i+1] << 16) | ((uint32_t)p[4*i+2] << 8) | p[4*i+3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i-15], 7) ^ rotr(w[i-15], 18) ^ (w[i-15] >> 3);
    uint32_t s1 = rotr(w[i-2], 17) ^ rotr(w[i-2], 19) ^ (w[i-2] >> 10);
    w[i] = w[i-16] + s0 + w[i-7] + s1;
  }
  uint32_t a=st[0],b=st[1],c=st[2],d=st[3],e=st[4],f=st[5],g=st[6],h=st[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = rotr(e,6) ^ rotr(e,11) ^ rotr(e,25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + K256[i] + w[i];
    uint32_t S0 = rotr(a,2) ^ rotr(a,13) ^ rotr(a,22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h=g; g=f; f=e; e=d+t1; d=c; c=b; b=a; a=t1+t2;
  }
  st[0]+=a; st[1]+=b; st[2]+=c; st[3]+=d; st[4]+=e; st[5]+=f; st[6]+=g; st[7]+=h;
}

/* mbedtls_sha256(data, len, hash, 0) — README.md:2543 */
void orc_sha256(const uint8_t* data, uint64_t len, uint8_t out[32]) {
  uint32_t st[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
  uint64_t full = len / 64;
  for (uint64_t i = 0; i < full; i++) sha256_block(st, data + 64 * i);
  uint8_t tail[128];
  uint64_t rem = len - 64 * full;
  memset(tail, 0, sizeof tail);
  if (rem) memcpy(tail, data + 64 * full, rem);
  tail[rem] = 0x80;
  int nb = (rem + 9 <= 64) ? 1 : 2;
  uint64_t bits = len * 8;
  for (int i = 0; i < 8; i++) tail[64 * nb - 1 - i] = (uint8_t)(bits >> (8 * i));
  sha256_block(st, tail);
  if (nb == 2) sha256_block(st, tail + 64);
  for (int i = 0; i < 8; i++) {
    out[4*i] = (uint8_t)(st[i] >> 24); out[4*i+1] = (uint8_t)(st[i] >> 16);
    out[4*i+2] = (uint8_t)(st[i] >> 8); out[4*i+3] = (uint8_t)st[i];
  }
}

void orc_sha256_chunks(const uint8_t* data, const uint64_t* cuts, uint64_t n_chunks, uint8_t* digests) {
  for (uint64_t i = 0; i < n_chunks; i++) orc_sha256(data + cuts[i], cuts[i+1] - cuts[i], digests + 32 * i);
}

/* L3 index semantics (README.md:1288-1292, 1542-1551): found -> refcount++ and pointer to the stored
 * chunk; new -> insert.  As a pure function of the digest array: first_occ[i] = smallest j with
 * digest[j] == digest[i]; refcount[j] = multiplicity at first occurrences. */
typedef struct { const uint8_t* d; uint64_t i; } dd_item;
static int dd_cmp(const void* a, const void* b) {
  const dd_item* x = (const dd_item*)a; const dd_item* y = (const dd_item*)b;
  int c = memcmp(x->d, y->d, 32);
  if (c) return c;
  return (x->i > y->i) - (x->i < y->i);
}
void orc_dedup(const uint8_t* digests, uint64_t n, uint64_t* first_occ, uint32_t* refcount) {
  if (n == 0) return;
  dd_item* it = (dd_item*)malloc(n * sizeof *it);
  for (uint64_t i = 0; i < n; i++) { it[i].d = digests + 32 * i; it[i].i = i; }
  qsort(it, n, sizeof *it, dd_cmp);
  for (uint64_t i = 0; i < n; i++) if (refcount) refcount[i] = 0;
  uint64_t r = 0;
  while (r < n) {
    uint64_t e = r + 1;
    while (e < n && memcmp(it[e].d, it[r].d, 32) == 0) e++;
    for (uint64_t k = r; k < e; k++) first_occ[it[k].i] = it[r].i;
    if (refcount) refcount[it[r].i] = (uint32_t)(e - r);
    r = e;
  }
  free(it);
}

/* ------------------------------------------------------------------ L4 */

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* MurmurHash3_x86_32 (public domain algorithm by A. Appleby; named by README.md:2573, 2591 but not
 * vendored by the reference). Pinned by the KAT vectors of SURVEY.md §8c. */
uint32_t orc_murmur3_x86_32(const void* key, int len, uint32_t seed) {
  const uint8_t* p = (const uint8_t*)key;
  uint32_t h = seed;
  const uint32_t c1 = 0xcc9e2d51, c2 = 0x1b873593;
  int nb = len / 4;
  for (int i = 0; i < nb; i++) {
    uint32_t k = (uint32_t)p[4*i] | ((uint32_t)p[4*i+1] << 8) | ((uint32_t)p[4*i+2] << 16) | ((uint32_t)p[4*i+3] << 24);
    k *= c1; k = rotl32(k, 15); k *= c2;
    h ^= k; h = rotl32(h, 13); h = h * 5 + 0xe6546b64;
  }
  const uint8_t* t = p + 4 * nb;
  uint32_t k = 0;
  switch (len & 3) {
    case 3: k ^= (uint32_t)t[2] << 16; /* fallthrough */
    case 2: k ^= (uint32_t)t[1] << 8;  /* fallthrough */
    case 1: k ^= t[0]; k *= c1; k = rotl32(k, 15); k *= c2; h ^= k;
  }
  h ^= (uint32_t)len;
  h ^= h >> 16; h *= 0x85ebca6b; h ^= h >> 13; h *= 0xc2b2ae35; h ^= h >> 16;
  return h;
}

/* minhash_compute (README.md:2578-2598): sig[h]=UINT32_MAX; for pos in [0,len-3): x=LE32(data+pos);
 * for h: v=murmur(x, 4, seed=h); sig[h]=min. len<4 -> all 0xFFFFFFFF (size_t underflow guarded). */
void orc_minhash(const uint8_t* data, uint64_t len, const hmse_cfg* cfg, uint32_t* sig) {
  uint32_t nh = cfg->n_hashes, sh = cfg->shingle;
  for (uint32_t h = 0; h < nh; h++) sig[h] = 0xFFFFFFFFu;
  if (len < sh) return;
  for (uint64_t pos = 0; pos + sh <= len; pos++) {
    for (uint32_t h = 0; h < nh; h++) {
      uint32_t v = orc_murmur3_x86_32(data + pos, (int)sh, cfg->seed_base + h);
      if (v < sig[h]) sig[h] = v;
    }
  }
}

void orc_minhash_chunks(const uint8_t* data, const uint64_t* cuts, const uint64_t* chunk_ids,
                        uint64_t n_sel, const hmse_cfg* cfg, uint32_t* sig) {
  for (uint64_t k = 0; k < n_sel; k++) {
    uint64_t c = chunk_ids ? chunk_ids[k] : k;
    orc_minhash(data + cuts[c], cuts[c+1] - cuts[c], cfg, sig + (uint64_t)cfg->n_hashes * k);
  }
}

/* LSH banding (README.md:1375-1383, 1987-1996): b bands of r rows; band key = murmur3 over the band's
 * r*4 little-endian bytes with seed = band index (the reference leaves the band hash unspecified);
 * two chunks are candidates iff some WHOLE band is equal; base = the earliest such chunk before i. */
typedef struct { const uint32_t* rows; uint64_t i; uint32_t r; } band_item;
static int band_cmp(const void* a, const void* b) {
  const band_item* x = (const band_item*)a; const band_item* y = (const band_item*)b;
  int c = memcmp(x->rows, y->rows, (size_t)x->r * 4);
  if (c) return c;
  return (x->i > y->i) - (x->i < y->i);
}
void orc_lsh(const uint32_t* sig, uint64_t n, const hmse_cfg* cfg, uint32_t* band_keys, int64_t* base) {
  uint32_t b = cfg->bands, r = cfg->rows, nh = cfg->n_hashes;
  for (uint64_t i = 0; i < n; i++) base[i] = -1;
  if (n == 0) return;
  band_item* it = (band_item*)malloc(n * sizeof *it);
  for (uint32_t bd = 0; bd < b; bd++) {
    for (uint64_t i = 0; i < n; i++) {
      const uint32_t* rows = sig + nh * i + (uint64_t)bd * r;
      if (band_keys) band_keys[i * b + bd] = orc_murmur3_x86_32(rows, (int)(r * 4), bd);
      it[i].rows = rows; it[i].i = i; it[i].r = r;
    }
    qsort(it, n, sizeof *it, band_cmp);
    uint64_t s = 0;
    while (s < n) {
      uint64_t e = s + 1;
      while (e < n && memcmp(it[e].rows, it[s].rows, (size_t)r * 4) == 0) e++;
      for (uint64_t k = s + 1; k < e; k++) {
        int64_t cand = (int64_t)it[s].i;
        int64_t* bp = &base[it[k].i];
        if (*bp < 0 || cand < *bp) *bp = cand;
      }
      s = e;
    }
  }
  free(it);
}
