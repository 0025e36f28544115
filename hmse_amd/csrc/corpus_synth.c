/*
 * corpus_synth.c — "wiki-synth(seed)" deterministic corpus generator (host code, pthreads).
 *
 * No corpus ships with the reference or this image (SURVEY.md §7 "No corpus in the container"),
 * so the bench and the tests run on this generator.  It follows the redundancy profile the
 * reference assumes for Wikipedia text (README.md:2123-2127: 15-20 % exact-duplicate,
 * 30-40 % similar-variant, 40-55 % unique) and the sampling seed of VALIDATION_METHODS.md:119-120.
 *
 * The corpus is a sequence of independent fixed-size blocks: block b depends only on
 * (seed, profile, b), so any rank can generate any block range (multi-GPU sharding) and a
 * prefix of the corpus is stable.  Inside a block, articles are appended until it is full:
 *   - "pool" articles are drawn from a fixed set of shared articles and re-emitted verbatim
 *     (exact duplicates at chunk granularity) or with sparse word edits (similar variants);
 *   - "fresh" articles are unique Zipf(1.1) prose over a seeded 50 k-word vocabulary.
 * Every article = title + infobox (one of 20 templates, parameter substitution) + paragraphs
 * + citations drawn from a shared pool.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VOCAB 50000u
#define ZTAB_BITS 20
#define NCITE 1000u
#define NTEMPL 20u
#define NPOOL 4096u

typedef struct {
  uint64_t seed;
  char* wtext;        /* all words, concatenated */
  uint32_t* woff;     /* VOCAB+1 */
  uint16_t* ztab;     /* 2^ZTAB_BITS quantile table -> word id */
  char* ctext;        /* citations */
  uint32_t* coff;     /* NCITE+1 */
  uint16_t tfield[NTEMPL][12];
} synth_t;

static synth_t* g_synth = 0;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static inline uint64_t sm64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t mix2(uint64_t a, uint64_t b) {
  uint64_t s = a ^ (b * 0xD6E8FEB86659FD93ull);
  return sm64(&s);
}

/* integer-only Zipf(1.1) quantile table: weight(r) ~ 1/(r+1)^1.1 computed with a fixed-point
 * pow so that the table is identical on every host (no libm dependence). */
static uint64_t fx_pow_1p1_inv(uint32_t r) {
  /* returns 2^40 / (r+1)^1.1 using exp2/log2 in 32.32 fixed point via repeated squaring */
  /* log2(x) for x = r+1 */
  uint64_t x = (uint64_t)(r + 1) << 32; /* 32.32 */
  int64_t ip = 0;
  while (x >= (2ull << 32)) { x >>= 1; ip++; }
  /* x in [1,2): fractional bits by squaring */
  uint64_t frac = 0;
  uint64_t y = x;
  for (int i = 0; i < 32; i++) {
    /* y = y*y (32.32) */
    unsigned __int128 t = (unsigned __int128)y * y;
    y = (uint64_t)(t >> 32);
    frac <<= 1;
    if (y >= (2ull << 32)) { y >>= 1; frac |= 1; }
  }
  int64_t lg = (ip << 32) | (int64_t)frac;   /* log2(r+1) in 32.32 */
  /* e = 40 - 1.1*lg */
  int64_t e = ((int64_t)40 << 32) - (lg + lg / 10);
  if (e < 0) return 0;
  int64_t ei = e >> 32; uint64_t ef = (uint64_t)e & 0xFFFFFFFFull;
  /* 2^ef via product of precomputed 2^(2^-k) in 2.62? use iterative sqrt-free method: */
  /* 2^f = prod over set bits k of 2^(2^-(k+1)); constants in 32.32 */
  static const uint64_t c[16] = {
    0x16A09E667ull, 0x1306FE0A3ull, 0x1172B83C7ull, 0x10B5586CFull, 0x1059B0D31ull, 0x102C9A3E7ull,
    0x10163DA9Full, 0x100B1AFA5ull, 0x10058C86Dull, 0x1002C605Eull, 0x100162F39ull, 0x1000B175Full,
    0x100058BA0ull, 0x10002C5CCull, 0x1000162E5ull, 0x10000B172ull};
  uint64_t m = 1ull << 32;
  for (int k = 0; k < 16; k++)
    if (ef & (1ull << (31 - k))) m = (uint64_t)(((unsigned __int128)m * c[k]) >> 32);
  return (uint64_t)(((unsigned __int128)m << ei) >> 32);
}

static synth_t* synth_init(uint64_t seed) {
  synth_t* S = (synth_t*)calloc(1, sizeof *S);
  S->seed = seed;
  uint64_t r = mix2(seed, 0x766F636162ull); /* "vocab" */
  S->woff = (uint32_t*)malloc((VOCAB + 1) * 4);
  S->wtext = (char*)malloc((size_t)VOCAB * 16);
  uint32_t off = 0;
  for (uint32_t w = 0; w < VOCAB; w++) {
    S->woff[w] = off;
    /* frequent words are short (1..4 letters for the top ranks), rare ones up to 13 */
    uint32_t lg = 0; for (uint32_t t = w + 2; t > 1; t >>= 1) lg++;
    uint32_t len = 1 + (uint32_t)(sm64(&r) % (2 + lg * 3 / 4));
    if (w >= 64 && len < 3) len += 2;
    for (uint32_t i = 0; i < len; i++) {
      uint64_t v = sm64(&r);
      /* letter distribution skewed like English */
      static const char L[] = "eeeeeeeeeeeetttttttttaaaaaaaaooooooooiiiiiiinnnnnnnsssssshhhhhhrrrrrrddddlllluuucccmmmwwffggyyppbbvkjxqz";
      S->wtext[off++] = L[v % (sizeof L - 1)];
    }
  }
  S->woff[VOCAB] = off;
  /* Zipf quantile table */
  uint64_t* cdf = (uint64_t*)malloc((size_t)VOCAB * 8);
  uint64_t tot = 0;
  for (uint32_t w = 0; w < VOCAB; w++) { tot += fx_pow_1p1_inv(w); cdf[w] = tot; }
  S->ztab = (uint16_t*)malloc(((size_t)1 << ZTAB_BITS) * 2);
  uint32_t w = 0;
  for (uint32_t i = 0; i < (1u << ZTAB_BITS); i++) {
    uint64_t target = (uint64_t)(((unsigned __int128)tot * i) >> ZTAB_BITS);
    while (w + 1 < VOCAB && cdf[w] <= target) w++;
    S->ztab[i] = (uint16_t)w;
  }
  free(cdf);
  /* citations */
  S->coff = (uint32_t*)malloc((NCITE + 1) * 4);
  S->ctext = (char*)malloc((size_t)NCITE * 256);
  uint64_t rc = mix2(seed, 0x63697465ull);
  off = 0;
  for (uint32_t c = 0; c < NCITE; c++) {
    S->coff[c] = off;
    S->ctext[off++] = '*'; S->ctext[off++] = ' ';
    uint32_t nw = 8 + (uint32_t)(sm64(&rc) % 12);
    for (uint32_t k = 0; k < nw; k++) {
      uint32_t wi = S->ztab[sm64(&rc) >> (64 - ZTAB_BITS)];
      if (k == 0 || k == 2) wi = 2000 + (uint32_t)(sm64(&rc) % 20000);
      uint32_t a = S->woff[wi], b = S->woff[wi + 1];
      memcpy(S->ctext + off, S->wtext + a, b - a);
      if (k < 3 && S->ctext[off] >= 'a') S->ctext[off] -= 32;
      off += b - a;
      S->ctext[off++] = (k == 1) ? ',' : ' ';
    }
    uint32_t yr = 1950 + (uint32_t)(sm64(&rc) % 75);
    S->ctext[off++] = '(';
    S->ctext[off++] = (char)('0' + yr / 1000); S->ctext[off++] = (char)('0' + yr / 100 % 10);
    S->ctext[off++] = (char)('0' + yr / 10 % 10); S->ctext[off++] = (char)('0' + yr % 10);
    S->ctext[off++] = ')'; S->ctext[off++] = '.'; S->ctext[off++] = '\n';
  }
  S->coff[NCITE] = off;
  uint64_t rt = mix2(seed, 0x74656D706Cull);
  for (uint32_t t = 0; t < NTEMPL; t++)
    for (int f = 0; f < 12; f++) S->tfield[t][f] = (uint16_t)(100 + sm64(&rt) % 3000);
  return S;
}

typedef struct { uint8_t* p; uint64_t cap, n; } sink_t;
static inline int sk_full(const sink_t* s) { return s->n >= s->cap; }
static inline void sk_put(sink_t* s, const void* src, uint32_t len) {
  uint64_t room = s->cap - s->n;
  if (len > room) len = (uint32_t)room;
  memcpy(s->p + s->n, src, len);
  s->n += len;
}
static inline void sk_c(sink_t* s, char c) { if (s->n < s->cap) s->p[s->n++] = (uint8_t)c; }

typedef struct { uint64_t r; uint64_t er; uint32_t rate; uint32_t prev; } ar_t; /* article rng, edit rng, edit rate (per 2^16), previous word */

/* Zipf draw with phrase structure: 5 times in 8 the word is one of 4 fixed successors of the
 * previous word, which gives the text the repeated bigrams/trigrams natural language has. */
static inline uint32_t next_word(const synth_t* S, ar_t* a) {
  uint64_t v = sm64(&a->r);
  uint32_t wi;
  if ((v & 7) < 5) wi = S->ztab[mix2(S->seed, ((uint64_t)a->prev << 2) | ((v >> 3) & 3)) >> (64 - ZTAB_BITS)];
  else wi = S->ztab[v >> (64 - ZTAB_BITS)];
  a->prev = wi;
  if (a->rate) {
    uint64_t e = sm64(&a->er);
    if ((e & 0xFFFF) < a->rate) wi = S->ztab[(e >> 16) & ((1u << ZTAB_BITS) - 1)];
  }
  return wi;
}
static inline void put_word(const synth_t* S, sink_t* s, uint32_t wi, int cap) {
  uint32_t a = S->woff[wi], b = S->woff[wi + 1];
  uint64_t at = s->n;
  sk_put(s, S->wtext + a, b - a);
  if (cap && at < s->cap && s->p[at] >= 'a') s->p[at] -= 32;
}

/* one article; structure is a function of aseed only, edits only swap words */
static void article(const synth_t* S, sink_t* s, uint64_t aseed, uint64_t eseed, uint32_t rate, uint32_t templ_share) {
  ar_t a; a.r = aseed; a.er = eseed; a.rate = rate; a.prev = 0;
  uint32_t target = 24576 + (uint32_t)(sm64(&a.r) % 73728); /* 24..96 KiB */
  uint64_t begin = s->n;
  sk_put(s, "= ", 2);
  uint32_t nt = 1 + (uint32_t)(sm64(&a.r) % 4);
  for (uint32_t k = 0; k < nt; k++) { put_word(S, s, 500 + (uint32_t)(sm64(&a.r) % 40000), 1); sk_c(s, k + 1 < nt ? ' ' : ' '); }
  sk_put(s, "=\n", 2);
  if (sm64(&a.r) % 100 < templ_share) {
    uint32_t t = (uint32_t)(sm64(&a.r) % NTEMPL);
    sk_put(s, "{{Infobox ", 10); put_word(S, s, S->tfield[t][0], 0); sk_c(s, '\n');
    for (int f = 1; f < 12; f++) {
      sk_put(s, "| ", 2); put_word(S, s, S->tfield[t][f], 0); sk_put(s, " = ", 3);
      uint32_t nv = 1 + (uint32_t)(sm64(&a.r) % 5);
      for (uint32_t k = 0; k < nv; k++) { put_word(S, s, next_word(S, &a), 0); sk_c(s, k + 1 < nv ? ' ' : '\n'); }
    }
    sk_put(s, "}}\n", 3);
  }
  while (s->n - begin < target && !sk_full(s)) {
    uint64_t v = sm64(&a.r);
    if (v % 16 == 0) { /* section heading */
      sk_put(s, "\n== ", 4); put_word(S, s, next_word(S, &a), 1); sk_put(s, " ==\n", 4);
    }
    uint32_t nsent = 3 + (uint32_t)((v >> 8) % 8);
    for (uint32_t q = 0; q < nsent; q++) {
      uint32_t nw = 6 + (uint32_t)(sm64(&a.r) % 22);
      for (uint32_t k = 0; k < nw; k++) {
        put_word(S, s, next_word(S, &a), k == 0);
        if (k + 1 < nw) { if ((sm64(&a.r) & 15) == 0) sk_c(s, ','); sk_c(s, ' '); }
      }
      sk_put(s, ". ", 2);
    }
    if ((v >> 20) % 4 == 0) { /* shared citation */
      uint32_t c = (uint32_t)(sm64(&a.r) % NCITE);
      sk_c(s, '\n'); sk_put(s, S->ctext + S->coff[c], S->coff[c + 1] - S->coff[c]);
    } else sk_c(s, '\n');
  }
  sk_c(s, '\n');
}

/* skewed pool draw (index ~ NPOOL * u^4) so that small corpora also contain repeats */
static inline uint64_t pool_pick(uint64_t x) {
  uint64_t u = x & 0xFFFF; /* u in [0, 2^16) */
  uint64_t u2 = (u * u) >> 16, u4 = (u2 * u2) >> 16;
  return (u4 * NPOOL) >> 16;
}

typedef struct { const synth_t* S; uint8_t* out; uint64_t b0, nb; uint32_t bs; int profile; int tid, nt; } job_t;

static void gen_block(const synth_t* S, uint8_t* out, uint64_t b, uint32_t bs, int profile) {
  /* profile: 0 wiki, 1 arxiv, 2 news, 3 code-ish */
  static const uint32_t pdup[4] = {18, 3, 10, 8}, pvar[4] = {35, 10, 25, 30}, tshare[4] = {80, 5, 30, 10};
  uint64_t r = mix2(S->seed ^ ((uint64_t)profile << 56), b + 1);
  sink_t s; s.p = out; s.cap = bs; s.n = 0;
  while (!sk_full(&s)) {
    uint32_t kind = (uint32_t)(sm64(&r) % 100);
    uint64_t x = sm64(&r);
    if (kind < pdup[profile]) {
      uint64_t pa = pool_pick(x);
      article(S, &s, mix2(S->seed, 0xA0000000ull + pa), 0, 0, tshare[profile]);
    } else if (kind < pdup[profile] + pvar[profile]) {
      uint64_t pa = pool_pick(x >> 8);
      article(S, &s, mix2(S->seed, 0xA0000000ull + pa), sm64(&r), 655 /* 1 % of words */, tshare[profile]);
    } else {
      article(S, &s, x ^ mix2(S->seed, b), 0, 0, tshare[profile]);
    }
  }
}

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  for (uint64_t k = (uint64_t)j->tid; k < j->nb; k += (uint64_t)j->nt)
    gen_block(j->S, j->out + k * j->bs, j->b0 + k, j->bs, j->profile);
  return 0;
}

/* Fill out[0 .. n_blocks*block_size) with blocks first_block.. of corpus (seed, profile). */
int hmse_corpus_generate(uint8_t* out, uint64_t first_block, uint64_t n_blocks, uint32_t block_size,
                         uint64_t seed, int profile, int n_threads) {
  if (!out || block_size < 4096 || profile < 0 || profile > 3) return -1;
  pthread_mutex_lock(&g_lock);
  if (!g_synth || g_synth->seed != seed) g_synth = synth_init(seed); /* tables are small; old ones leak by design */
  const synth_t* S = g_synth;
  pthread_mutex_unlock(&g_lock);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  if ((uint64_t)n_threads > n_blocks) n_threads = (int)(n_blocks ? n_blocks : 1);
  pthread_t th[64]; job_t jobs[64];
  for (int t = 0; t < n_threads; t++) {
    jobs[t].S = S; jobs[t].out = out; jobs[t].b0 = first_block; jobs[t].nb = n_blocks; jobs[t].bs = block_size;
    jobs[t].profile = profile; jobs[t].tid = t; jobs[t].nt = n_threads;
    pthread_create(&th[t], 0, worker, &jobs[t]);
  }
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], 0);
  return 0;
}
