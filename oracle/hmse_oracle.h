/*
 * hmse_oracle.h — CPU ORACLE for the HMSE L1-L4 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (hmse_amd/) never links, imports or calls anything under oracle/.
 *
 * Parity status: the reference (1Jamie/HMSE) ships NO implementation
 * and NO golden vectors for this path.
 *   - L3 SHA-256: pinned by FIPS 180-4 / hashlib / NIST vectors (tests/test_oracle.py).
 *   - L4a MurmurHash3_x86_32: pinned by the known-answer vectors of SURVEY.md §8c.
 *   - L1 DEFLATE: pinned by RFC 1951 decodability through stock zlib (round trip).
 *   - L1 inflate (read path): pinned against stock zlib 1.2.11 — same bytes on valid streams, same
 *     accept/reject decision on mutated ones (tests/test_oracle.py).
 *   - L2 Gear-FastCDC cut points, L4 signatures/LSH bases: PARITY UNPINNED at the
 *     reference; pinned here by an independent pure-Python restatement
 *     (oracle/pyref.py) and committed fixtures (tests/golden/).
 */
#ifndef HMSE_ORACLE_H
#define HMSE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/hmse.h" /* hmse_cfg layout + status codes only */

#ifdef __cplusplus
extern "C" {
#endif

void     orc_cfg_default(hmse_cfg* cfg);
void     orc_gear_table(uint64_t table[256]);
void     orc_cdc_masks(const hmse_cfg* cfg, uint64_t* mask_s, uint64_t* mask_l);

/* L2: returns number of chunks; cuts[0]=0, cuts[1..n] chunk END offsets. cuts may be NULL (count only). */
uint64_t orc_cdc(const uint8_t* data, uint64_t n, const uint64_t* seg_off, uint32_t n_seg,
                 const hmse_cfg* cfg, uint64_t* cuts, uint64_t cuts_cap);
/* the literal reference skeleton (README.md:2456-2490), for the acceptance-band test */
uint64_t orc_cdc_reference_skeleton(const uint8_t* data, uint64_t n, uint64_t* cuts, uint64_t cuts_cap);

/* L3 */
void     orc_sha256(const uint8_t* data, uint64_t len, uint8_t out[32]);
void     orc_sha256_chunks(const uint8_t* data, const uint64_t* cuts, uint64_t n_chunks, uint8_t* digests);
void     orc_dedup(const uint8_t* digests, uint64_t n, uint64_t* first_occ, uint32_t* refcount);

/* L4 */
uint32_t orc_murmur3_x86_32(const void* key, int len, uint32_t seed);
void     orc_minhash(const uint8_t* data, uint64_t len, const hmse_cfg* cfg, uint32_t* sig);
void     orc_minhash_chunks(const uint8_t* data, const uint64_t* cuts, const uint64_t* chunk_ids,
                            uint64_t n_sel, const hmse_cfg* cfg, uint32_t* sig);
void     orc_lsh(const uint32_t* sig, uint64_t n_sel, const hmse_cfg* cfg, uint32_t* band_keys, int64_t* base);

/* L1: raw DEFLATE of chunk[0..len) with optional preset dictionary; returns stream bytes or <0 */
int64_t  orc_deflate(const uint8_t* chunk, uint32_t len, const uint8_t* dict, uint32_t dict_len,
                     const hmse_cfg* cfg, uint8_t* out, uint64_t out_cap);
uint32_t orc_deflate_bound(uint32_t len);
/* batch form mirroring hmse_l1_deflate (dense out, out_off, kind) */
int      orc_deflate_chunks(const uint8_t* data, const uint64_t* cuts, const uint64_t* chunk_ids,
                            const int64_t* base, uint64_t n_sel, const hmse_cfg* cfg, uint8_t* out,
                            uint64_t out_cap, uint64_t* out_off, uint8_t* kind);
/* debug hooks used by the parity tests to localise a mismatch */
void     orc_deflate_matches(const uint8_t* chunk, uint32_t len, const uint8_t* dict, uint32_t dict_len,
                             const hmse_cfg* cfg, uint16_t* mlen, uint16_t* mdist);

/* read path: raw-DEFLATE decode (RFC 1951; rejected set = stock zlib's, see hmse_oracle_inflate.c); 0 or <0 reason */
int      orc_inflate(const uint8_t* stream, uint64_t stream_len, const uint8_t* dict, uint32_t dict_len,
                     uint8_t* out, uint32_t raw_len);
/* batch form mirroring hmse_l1_inflate; returns the number of corrupt streams, ok[k] = 1 where chunk k decoded */
uint64_t orc_inflate_chunks(const uint8_t* streams, const uint64_t* stream_off, const uint32_t* stream_len,
                            const uint8_t* kind, const int64_t* base, uint64_t n_sel, const uint64_t* raw_off,
                            uint8_t* raw_out, uint8_t* ok);

#ifdef __cplusplus
}
#endif
#endif
