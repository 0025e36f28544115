/*
 * lz_study.c — STUDY HARNESS (test infrastructure; includes the CPU oracle's translation unit, so it lives beside the
 * oracle's users: tools/ + tests only, never hmse_amd/).  VERDICT r3 item 1: "positions that are never walked".
 *
 * study_deflate() encodes one chunk (optionally against a dictionary) under a VARIANT of the matcher and returns the stream size and
 * work counters, so that definitions can be compared on stored bytes (vs the current rule, vs zlib-9) and on
 * walks/candidates BEFORE any kernel is touched.  The encoder (rules 3-6) is the oracle's own.
 *
 *   params[0] D      depth of a full walk (32 = level 9)
 *   params[1] mode   0 = every position walked (the round-3 definition)
 *                    1 = pass 1 (depth K1, length cap C1) everywhere -> parse -> full walks only at the positions the
 *                        parse READS (p and p+1 of every step) -> parse again, `iters` times (0 = until nothing new)
 *   params[2] K1     pass-1 depth
 *   params[3] C1     pass-1 length cap (bytes compared)
 *   params[4] iters
 *   stats[0] chunk positions   [1] positions fully walked   [2] candidates looked at by full walks
 *        [3] candidates looked at by pass 1   [4] tokens   [5] iterations used   [6] positions the final parse read
 *        [7] positions read by the final parse that were NOT fully walked
 */
#include "../../oracle/hmse_oracle_deflate.c"

typedef struct { uint32_t T, Dl, nh; uint8_t* W; uint32_t* start; uint16_t* S; uint32_t* rank; } idx_t;

static void idx_build(idx_t* x, const uint8_t* chunk, uint32_t len, const uint8_t* dict, uint32_t dlen) {
  if (dlen > WMAX) { dict += dlen - WMAX; dlen = WMAX; }
  uint32_t T = dlen + len;
  x->T = T; x->Dl = dlen;
  x->W = (uint8_t*)malloc(T + 64);
  if (dlen) memcpy(x->W, dict, dlen);
  memcpy(x->W + dlen, chunk, len); memset(x->W + T, 0, 64);
  uint32_t nh = T >= 4 ? T - 3 : 0; x->nh = nh;
  x->start = (uint32_t*)calloc(NBUCKET + 1, 4);
  uint32_t* fill = (uint32_t*)calloc(NBUCKET, 4);
  x->S = (uint16_t*)malloc((nh + 1) * 2); x->rank = (uint32_t*)malloc((nh + 1) * 4);
  for (uint32_t q = 0; q < nh; q++) x->start[hash4(le32(x->W + q)) + 1]++;
  for (uint32_t h = 0; h < NBUCKET; h++) x->start[h + 1] += x->start[h];
  for (uint32_t q = 0; q < nh; q++) { uint32_t h = hash4(le32(x->W + q)); uint32_t r = x->start[h] + fill[h]++; x->S[r] = (uint16_t)q; x->rank[q] = r; }
  free(fill);
}
static void idx_free(idx_t* x) { free(x->W); free(x->start); free(x->S); free(x->rank); }

/* rule 2 at position p with depth D and length cap C; returns candidates looked at */
static uint32_t walk(const idx_t* x, uint32_t p, uint32_t D, uint32_t C, uint16_t* ml_out, uint16_t* md_out) {
  *ml_out = 0; *md_out = 0;
  if (p + 4 > x->T) return 0;
  uint32_t h = hash4(le32(x->W + p)), r = x->rank[p], g = x->start[h];
  uint32_t maxlen = x->T - p < MAXM ? x->T - p : MAXM;
  if (maxlen > C) maxlen = C;
  uint32_t best = MINM - 1, bdist = 0, n = 0;
  for (uint32_t k = 1; k <= D && r >= g + k; k++) {
    uint32_t q = x->S[r - k];
    if (p - q > WMAX) break;
    n++;
    uint32_t ml = 0;
    while (ml < maxlen && x->W[q + ml] == x->W[p + ml]) ml++;
    if (ml > best) { best = ml; bdist = p - q; if (ml == maxlen) break; }
  }
  if (best >= MINM) { *ml_out = (uint16_t)best; *md_out = (uint16_t)bdist; }
  return n;
}

/* the parse of rule 3, marking what it reads; returns tokens */
static uint32_t parse_reads(const uint16_t* ml, uint32_t len, uint8_t* rd) {
  memset(rd, 0, len + 1);
  uint32_t nt = 0;
  for (uint32_t p = 0; p < len;) {
    uint32_t m = ml[p]; rd[p] = 1;
    if (m >= MINM) { if (p + 1 < len) rd[p + 1] = 1; }
    if (m >= MINM && !(p + 1 < len && ml[p + 1] > m)) p += m; else p++;
    nt++;
  }
  return nt;
}

int64_t study_deflate(const uint8_t* chunk, uint32_t len, const uint8_t* dict, uint32_t dlen, const uint32_t* params,
                      uint64_t* stats, uint8_t* out, uint64_t out_cap) {
  uint32_t D = params[0], mode = params[1], K1 = params[2], C1 = params[3], iters = params[4];
  memset(stats, 0, 8 * 8);
  if (dlen) return -1; /* plain jobs only so far */
  idx_t x; idx_build(&x, chunk, len, dict, dlen);
  uint16_t* ml = (uint16_t*)calloc(len + 2, 2); uint16_t* md = (uint16_t*)calloc(len + 2, 2);
  uint8_t* walked = (uint8_t*)calloc(len + 2, 1); uint8_t* rd = (uint8_t*)calloc(len + 2, 1);
  stats[0] = len;
  if (mode == 0) {
    for (uint32_t p = 0; p < len; p++) { stats[2] += walk(&x, p, D, MAXM, &ml[p], &md[p]); walked[p] = 1; }
    stats[1] = len; stats[5] = 1;
  } else {
    for (uint32_t p = 0; p < len; p++) stats[3] += walk(&x, p, K1, C1, &ml[p], &md[p]);
    for (uint32_t it = 0; iters == 0 || it < iters; it++) {
      parse_reads(ml, len, rd);
      uint32_t fresh = 0;
      for (uint32_t p = 0; p < len; p++) if (rd[p] && !walked[p]) {
        stats[2] += walk(&x, p, D, MAXM, &ml[p], &md[p]); walked[p] = 1; fresh++; }
      stats[1] += fresh; stats[5]++;
      if (!fresh) break;
    }
  }
  stats[4] = parse_reads(ml, len, rd);
  for (uint32_t p = 0; p < len; p++) { stats[6] += rd[p]; stats[7] += rd[p] && !walked[p]; }
  int64_t n = encode_matches(chunk, len, ml, md, out, out_cap);
  free(ml); free(md); free(walked); free(rd); idx_free(&x);
  return n;
}

/* ---- trip model of the kernel's state machine (exact walks; only the COST of a walk is modelled) ----------------------------
 * A lane makes one trip per unit of work.  Filter variants decide which candidates are consumed without a window read and how
 * many of those can be consumed per trip (G = filter entries examined per trip; the first that passes is probed in that trip).
 *   fmode 0: round-3 kernel: K = (byte4 & 15) | 4 further bits of the hash product; G = 2
 *   fmode 1: 16-bit entry: 4 hash bits | low nibbles of bytes 4, 5, 6
 *   fmode 2: 8-bit entry: low nibbles of bytes 4 and 5 (no hash bits)
 * out[0] probe trips, out[1] extend trips, out[2] candidates, out[3] candidates consumed by the filter, out[4] window probes */
static int filt_rejects(const idx_t* x, uint32_t p, uint32_t q, uint32_t best, int fmode) {
  const uint8_t* W = x->W;
  uint32_t hp = le32(W + p) * 0x9E3779B1u, hq = le32(W + q) * 0x9E3779B1u;
  int hashdiff = ((hp >> 16) & 0xF) != ((hq >> 16) & 0xF);
  if (fmode == 0) return hashdiff || (best >= 4 && ((W[p + 4] ^ W[q + 4]) & 15));
  if (fmode == 1) {
    if (hashdiff) return 1;
    for (uint32_t j = 0; j < 3; j++) if (best >= 4 + j && ((W[p + 4 + j] ^ W[q + 4 + j]) & 15)) return 1;
    return 0;
  }
  if (fmode == 3) {   /* K8 + the two spare bits of a 14-bit S entry: low 2 bits of byte 5 */
    if (hashdiff || (best >= 4 && ((W[p + 4] ^ W[q + 4]) & 15))) return 1;
    return best >= 5 && ((W[p + 5] ^ W[q + 5]) & 3);
  }
  if (fmode == 4) {   /* 6 hash bits + low nibble of byte 4 (two spare bits of S spent on the hash) */
    return ((hp >> 14) & 0x3F) != ((hq >> 14) & 0x3F) || (best >= 4 && ((W[p + 4] ^ W[q + 4]) & 15));
  }
  if (fmode == 5) {   /* 4 hash bits + low 6 bits of byte 4 */
    return hashdiff || (best >= 4 && ((W[p + 4] ^ W[q + 4]) & 63));
  }
  if (fmode == 6) {   /* 12 bits: 4 hash bits + nibbles of bytes 4, 5 */
    if (hashdiff) return 1;
    for (uint32_t j = 0; j < 2; j++) if (best >= 4 + j && ((W[p + 4 + j] ^ W[q + 4 + j]) & 15)) return 1;
    return 0;
  }
  for (uint32_t j = 0; j < 2; j++) if (best >= 4 + j && ((W[p + 4 + j] ^ W[q + 4 + j]) & 15)) return 1;
  return 0;
}
static void walk_trips(const idx_t* x, uint32_t p, uint32_t D, int fmode, uint32_t G, uint64_t* out) {
  if (p + 4 > x->T) return;
  uint32_t h = hash4(le32(x->W + p)), r = x->rank[p], g = x->start[h];
  uint32_t maxlen = x->T - p < MAXM ? x->T - p : MAXM;
  uint32_t kmax = r - g < D ? r - g : D, best = MINM - 1, kk = 1;
  while (kk <= kmax) {
    out[0]++;
    uint32_t seen = 0, q = 0; int pass = 0;
    while (kk <= kmax && seen < G) {          /* up to G filter entries in this trip */
      q = x->S[r - kk]; seen++; out[2]++;
      if (p - q > WMAX) { kk = kmax + 1; break; }
      if (!filt_rejects(x, p, q, best, fmode)) { pass = 1; break; }
      out[3]++; kk++;
    }
    if (!pass) continue;
    out[4]++;
    uint32_t ml = 0;
    while (ml < maxlen && x->W[q + ml] == x->W[p + ml]) ml++;
    if (ml >= 8 && maxlen > 8) { uint32_t e = 8; do { out[1]++; e += 32; } while (e <= ml && e < maxlen);
      e = 8; do { out[5]++; e += 16; } while (e <= ml && e < maxlen); e = 8; do { out[6]++; e += 8; } while (e <= ml && e < maxlen);
      out[7] += ml < 16; }
    if (ml > best) { best = ml; if (ml == maxlen) break; }
    kk++;
  }
}
void study_trips(const uint8_t* chunk, uint32_t len, uint32_t D, int fmode, uint32_t G, const uint8_t* only, uint64_t* out) {
  idx_t x; idx_build(&x, chunk, len, NULL, 0);
  for (uint32_t p = 0; p < len; p++) if (!only || only[p]) walk_trips(&x, p, D, fmode, G, out);
  idx_free(&x);
}

/* ---- scheduling model of the plain-job state machine -------------------------------------------------------------------------
 * NW wavefronts of 64 lanes share one queue of sorted ranks.  A lane's walk of rank r lasts trips[r] trips (model of walk_trips,
 * fmode 0, G 2).  Per trip of a wavefront: pull if >= batch lanes are idle (or no lane is walking); every walking lane advances one
 * trip.  variant bit 0: every lane holds a PREFETCHED next rank (refilled by the batched pull), so a lane that finishes starts its
 * next walk in the same trip; bit 1: cooperative tail — once the queue is empty and <= ctail lanes of a wavefront still walk, each
 * of them is finished by the whole wavefront in one trip.
 * out[0] wave-trips  [1] pulls executed  [2] lane-trips walking  [3] lane-trips idle (waiting for a pull)  [4] lane-trips done */
void study_sched(const uint8_t* chunk, uint32_t len, uint32_t D, uint32_t NW, uint32_t batch, uint32_t variant, uint32_t ctail, uint64_t* out) {
  idx_t x; idx_build(&x, chunk, len, NULL, 0);
  uint32_t nh = x.nh;
  uint32_t* trips = (uint32_t*)calloc(nh + 1, 4);
  for (uint32_t r = 0; r < nh; r++) { uint64_t o[8] = {0}; walk_trips(&x, x.S[r], D, 0, 2, o); trips[r] = (uint32_t)(o[0] + ((batch >> 8) == 1 ? 0 : (batch >> 8) == 2 ? (o[1] + 1) / 2 : o[1])); }
  batch &= 255;
  uint32_t nh_drop = 0;
  /* hand-out order: (variant >> 4) 0 = rank order; 1 = perfect longest-first (by modelled trips); 2 = chains >= othr candidates first, then the
     rest (two passes over the ranks); 3 = four classes by chain length (>= 24, >= 16, >= 8, rest) */
  {
    uint32_t om = variant >> 4, othr = ctail >> 8;
    if (om && om < 4) {
      uint32_t* key = (uint32_t*)calloc(nh + 1, 4); uint32_t* t2 = (uint32_t*)calloc(nh + 1, 4);
      for (uint32_t r = 0; r < nh; r++) {
        uint32_t h = hash4(le32(x.W + x.S[r])), km = r - x.start[h]; if (km > D) km = D;
        key[r] = om == 1 ? trips[r] : om == 2 ? (km >= othr) : othr == 97 ? (km >= 24 ? 4 : km >= 12 ? 3 : km >= 4 ? 2 : km >= 1 ? 1 : 0) : othr == 96 ? (km >= 16 ? 3 : km >= 4 ? 2 : km >= 1 ? 1 : 0) : othr == 95 ? (km >= 1) : othr == 99 ? km : othr == 98 ? (km >= 32 ? 7 : km >= 24 ? 6 : km >= 16 ? 5 : km >= 12 ? 4 : km >= 8 ? 3 : km >= 4 ? 2 : km >= 2 ? 1 : 0) : (km >= 24 ? 3 : km >= 16 ? 2 : km >= 8 ? 1 : 0);
      }
      uint32_t n = 0;
      for (int k = 1024; k >= 0; k--) for (uint32_t r = 0; r < nh; r++) if (key[r] == (uint32_t)k || (k == 1024 && key[r] > 1024)) t2[n++] = trips[r];
      if (om == 3 && othr >= 95 && othr <= 98) { uint32_t n0 = 0; for (uint32_t r = 0; r < nh; r++) n0 += key[r] == 0; n -= n0; for (uint32_t r = n; r < nh; r++) t2[r] = 0; nh_drop = n0; }
      memcpy(trips, t2, nh * 4); free(key); free(t2);
    }
    if (om >= 4) {
      /* bucket-level layouts: buckets grouped by SIZE class (hash order inside a class), members in bucket order; the queue runs from the
         deepest member of the biggest class downwards.  om 4: two classes (size >= othr | rest); om 5: four classes (>= 32, >= 16, >= 8, rest) */
      uint32_t* t2 = (uint32_t*)calloc(nh + 1, 4); uint32_t n = 0;
      int ncls = om == 4 ? 2 : 4;
      for (int c = ncls - 1; c >= 0; c--) {
        uint32_t seg0 = n;
        for (uint32_t h = 0; h < NBUCKET; h++) {
          uint32_t sz = x.start[h + 1] - x.start[h];
          int cl = om == 4 ? (sz >= othr) : (sz >= 32 ? 3 : sz >= 16 ? 2 : sz >= 8 ? 1 : 0);
          if (cl != c) continue;
          for (uint32_t r = x.start[h]; r < x.start[h + 1]; r++) t2[n++] = trips[r];
        }
        /* reverse this class's run: deepest members of its last bucket first */
        for (uint32_t a = seg0, b = n; a + 1 < b; a++, b--) { uint32_t tmp = t2[a]; t2[a] = t2[b - 1]; t2[b - 1] = tmp; }
      }
      memcpy(trips, t2, nh * 4); free(t2);
    }
    variant &= 15; ctail &= 255;
  }
  nh -= nh_drop;
  uint32_t qhead = 0;
  typedef struct { uint32_t rem[64]; uint8_t st[64]; uint32_t nxt[64]; uint8_t has[64]; int fin; } wv_t;  /* st: 0 idle, 1 walking, 2 done */
  wv_t* w = (wv_t*)calloc(NW, sizeof(wv_t));
  int pf = variant & 1, coop = (variant >> 1) & 1;
  uint32_t live = NW;
  while (live) {
    for (uint32_t v = 0; v < NW; v++) {
      wv_t* W = &w[v];
      if (W->fin) continue;
      out[0]++;
      uint32_t nidle = 0, nwalk = 0, nempty = 0;
      for (int l = 0; l < 64; l++) { nidle += W->st[l] == 0; nwalk += W->st[l] == 1; nempty += pf && !W->has[l] && W->st[l] != 2; }
      if (!pf) {
        if (nidle && (nidle >= batch || nwalk == 0)) {
          out[1]++;
          for (int l = 0; l < 64; l++) if (W->st[l] == 0) {
            if (qhead >= nh) W->st[l] = 2;
            else { uint32_t r = qhead++; if (trips[r]) { W->st[l] = 1; W->rem[l] = trips[r]; } }
          }
        }
      } else {
        if (nempty && (nempty >= batch || nwalk == 0)) {
          out[1]++;
          for (int l = 0; l < 64; l++) if (!W->has[l] && W->st[l] != 2) {
            while (qhead < nh && trips[qhead] == 0) qhead++;      /* (first-of-bucket ranks need no walk; the pull skips them) */
            if (qhead < nh) { W->nxt[l] = qhead++; W->has[l] = 1; }
          }
        }
        for (int l = 0; l < 64; l++) if (W->st[l] == 0) {
          if (W->has[l]) { W->st[l] = 1; W->rem[l] = trips[W->nxt[l]]; W->has[l] = 0; }
          else if (qhead >= nh) W->st[l] = 2;
        }
      }
      nwalk = 0; for (int l = 0; l < 64; l++) nwalk += W->st[l] == 1;
      if (coop && qhead >= nh && nwalk && nwalk <= ctail) {
        /* one lane's remaining walk per trip, done by all lanes */
        for (int l = 0; l < 64; l++) if (W->st[l] == 1) { W->rem[l] = 1; break; }
        int first = 1;
        for (int l = 0; l < 64; l++) if (W->st[l] == 1) { if (first) { first = 0; } else { out[3]++; continue; } }
      }
      int coop_now = coop && qhead >= nh && nwalk && nwalk <= ctail;
      int advanced = 0;
      for (int l = 0; l < 64; l++) {
        if (W->st[l] == 1) {
          if (coop_now && advanced) continue;   /* the others wait their turn */
          advanced = 1; out[2]++;
          if (--W->rem[l] == 0) W->st[l] = (pf && !W->has[l] && qhead >= nh) ? 2 : 0;
        } else if (W->st[l] == 0) out[3]++; else out[4]++;
      }
      int any = 0; for (int l = 0; l < 64; l++) any |= W->st[l] != 2;
      if (!any) { W->fin = 1; live--; }
    }
  }
  free(w); free(trips); idx_free(&x);
}

/* histogram of modelled walk lengths: hist[min(trips,63)] += 1, and by chain length: byk[min(kmax,32)][0..1] = walks, trips */
void study_walk_hist(const uint8_t* chunk, uint32_t len, uint32_t D, uint64_t* hist, uint64_t* byk) {
  idx_t x; idx_build(&x, chunk, len, NULL, 0);
  for (uint32_t r = 0; r < x.nh; r++) {
    uint64_t o[8] = {0}; walk_trips(&x, x.S[r], D, 0, 2, o);
    uint32_t t = (uint32_t)(o[0] + o[1]); hist[t > 63 ? 63 : t]++;
    uint32_t h = hash4(le32(x.W + x.S[r])), km = r - x.start[h]; if (km > D) km = D;
    byk[2 * km]++; byk[2 * km + 1] += t;
  }
  idx_free(&x);
}
