/*
 * hmse.h — C-ABI of the MI355X-native HMSE ingest hot path (L2 -> L3 -> L4 -> L1).
 *
 * This is the drop-in boundary.  The reference (1Jamie/HMSE, README.md) defines no
 * plugin/FFI interface: it has single-buffer C signatures with caller-owned output
 * and status-int returns.  Each entry point below is the batch form
 * of one of those signatures and cites the reference lines it replaces.
 *
 * Conventions (all entry points):
 *   - every pointer marked DEVICE is a HIP device pointer on the current device;
 *     the caller owns every buffer; the library never allocates and never syncs
 *     (hipGraph-capturable), work is stream-ordered on `stream` (a hipStream_t
 *     passed as void* so this header needs no HIP include);
 *   - return 0 = HMSE_OK, <0 = HMSE_E*;  counts that are only known on the device
 *     (n_cuts, overflow flags, stream lengths) are written to DEVICE memory;
 *   - one stream = one engine thread (README.md:148-153, "Core 1 HMSE engine").
 *   - integer/byte results are bit-exact against oracle/ (tests/).
 */
#ifndef HMSE_H
#define HMSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): contracts of existing entry points changed in round 3 — hmse_stream_batch / hmse_stream_piece_* need a workspace
 * prepared by hmse_stream_workspace_init (else status bit4) and take the global chunk count from state[8] (one rank: state[8] must
 * equal state[1]; a mismatch is status bit5); hmse_l1_deflate keeps ONE record per chunk in its workspace
 * (hmse_l1_deflate_record_bytes / _dict).  A caller built against version 1 must check hmse_abi_version() and refuse. */
#define HMSE_ABI_VERSION 2

enum {
  HMSE_OK      = 0,
  HMSE_EINVAL  = -1, /* bad argument / unsupported configuration              */
  HMSE_ENOSPC  = -2, /* a caller-provided buffer or workspace is too small     */
  HMSE_EHIP    = -3  /* a HIP runtime call failed (launch error)               */
};

/* stage ids for hmse_workspace_bytes() */
enum {
  HMSE_STAGE_L2_CDC     = 2,
  HMSE_STAGE_L3_SHA256  = 3,
  HMSE_STAGE_L3_DEDUP   = 4,
  HMSE_STAGE_L4_MINHASH = 5,
  HMSE_STAGE_L4_LSH     = 6,
  HMSE_STAGE_L1_DEFLATE = 7,
  /* read path; ids 8..15 are the DEFLATE kernels' profiling slots */
  HMSE_STAGE_L1_INFLATE    = 16,
  HMSE_STAGE_READ_ASSEMBLE = 17,
  HMSE_STAGE_MANIFEST_PACK = 18
};

/* layer-enable mask == the reference's ablation matrix / degradation modes
 * (VALIDATION_METHODS.md:458-464, README.md:745-770).                          */
enum {
  HMSE_LAYER_L1 = 1u, /* DEFLATE                */
  HMSE_LAYER_L2 = 2u, /* content-defined chunks */
  HMSE_LAYER_L3 = 4u, /* SHA-256 exact dedupe   */
  HMSE_LAYER_L4 = 8u  /* MinHash/LSH + delta    */
};

/* chunk kinds in the manifest (README.md:1635-1669) */
enum { HMSE_KIND_FULL = 0, HMSE_KIND_POINTER = 1, HMSE_KIND_DELTA = 2 };

/*
 * Configuration: mirrors the reference's compile-time constants
 * (README.md:2354-2355, 2444-2447, 2575-2576; SURVEY.md §5 "Config / flags").
 */
typedef struct hmse_cfg {
  uint32_t struct_size;  /* sizeof(hmse_cfg), ABI check                               */
  /* L2 — FASTCDC_MIN/AVG/MAX_SIZE (README.md:2444-2446), defaults 2048/8192/32768    */
  uint32_t min_size;     /* >= 64 (the Gear window)                                   */
  uint32_t avg_size;     /* power of two                                              */
  uint32_t max_size;     /* <= 32768 (uint16_t length in ChunkIndex, README.md:1267)  */
  uint32_t norm_level;   /* FastCDC normalisation: masks use log2(avg) +/- norm bits  */
  uint32_t seg_size;     /* resolve restarts every seg_size bytes (default 4 MiB)     */
  /* L4 — NUM_HASHES (README.md:2575), 4-byte shingles (README.md:2584-2586)          */
  uint32_t n_hashes;     /* 128                                                       */
  uint32_t shingle;      /* 4                                                         */
  uint32_t seed_base;    /* seeds are seed_base + 0..n_hashes-1 (README.md:2589-2591) */
  uint32_t bands;        /* b = 4  (README.md:1987-1996)                              */
  uint32_t rows;         /* r = 32, bands*rows == n_hashes                            */
  uint32_t band_bits;    /* 16 -> 65536 buckets per band (README.md:1937-1945)        */
  /* L1 — mz_deflateInit2(&s, 9, MZ_DEFLATED, 15, 9, ...) (README.md:2374)            */
  uint32_t level;        /* 1..9 profile; selects chain_depth when that is 0          */
  uint32_t chain_depth;  /* candidates examined per position (0 = from level)         */
  uint32_t layers;       /* HMSE_LAYER_* mask                                         */
  uint32_t delta_max_ratio_pct; /* optional gate: delta <= pct% of chunk (0 = off)    */
} hmse_cfg;

/* Fill *cfg with the defaults above. */
void hmse_cfg_default(hmse_cfg* cfg);
/* 0 if the configuration is supported by the device path, HMSE_EINVAL otherwise. */
int hmse_cfg_validate(const hmse_cfg* cfg);

int hmse_abi_version(void);
const char* hmse_strerror(int code);

/* The 256-entry Gear table (host copy), generated from a fixed seed. */
void hmse_gear_table(uint64_t table[256]);

/* Bytes of DEVICE workspace stage `stage` needs for an input of n bytes
 * (L2) or n chunks (all other stages). */
size_t hmse_workspace_bytes(int stage, uint64_t n, const hmse_cfg* cfg);

/*
 * L2 — content-defined chunking.  Replaces rabin_slide() + the cut loop of
 * benchmark_fastcdc() (README.md:2456-2464, 2475-2490): rolling state never reset at
 * a cut, cut iff size >= MIN && (hash hit || size >= MAX); Gear roll + two-mask
 * normalisation; resolution restarts at every segment boundary.
 *   data     DEVICE u8[n]
 *   seg_off  DEVICE u64[n_seg+1], ascending, seg_off[0]=0, seg_off[n_seg]=n
 *   cuts     DEVICE u64[cuts_cap]: receives cuts[0]=0 and then every chunk END offset
 *   n_cuts   DEVICE u64[1]: number of chunks (cuts holds n_cuts+1 entries).
 *            If the required count exceeds cuts_cap-1 nothing past cuts_cap is
 *            written and n_cuts still receives the required count.
 *   status   DEVICE u32[1]: 0 ok, bit0 = candidate workspace overflow,
 *            bit1 = cuts_cap overflow
 */
int hmse_l2_cdc(const uint8_t* data, uint64_t n, const uint64_t* seg_off, uint32_t n_seg,
                const hmse_cfg* cfg, uint64_t* cuts, uint64_t cuts_cap, uint64_t* n_cuts,
                uint32_t* status, void* ws, size_t ws_bytes, void* stream);

/*
 * L3 — SHA-256 of every chunk.  Replaces mbedtls_sha256(data, len, hash, 0)
 * (README.md:2543), applied to the raw chunk.
 *   cuts     DEVICE u64[n_chunks+1]
 *   digests  DEVICE u8[n_chunks][32]
 */
int hmse_l3_sha256(const uint8_t* data, uint64_t n, const uint64_t* cuts, uint64_t n_chunks,
                   uint8_t* digests, void* ws, size_t ws_bytes, void* stream);

/*
 * L3 — exact dedupe.  Replaces the hash-index probe/insert of README.md:1288-1292,
 * 1542-1551 on a table held whole in HBM.
 *   first_occ DEVICE u64[n_all]: index of the earliest chunk with an equal digest
 *   refcount  DEVICE u32[n_all]: at a first occurrence, number of chunks equal to it
 *             (ChunkIndex.refcount, README.md:1269); 0 elsewhere
 */
int hmse_l3_dedup(const uint8_t* digests_all, uint64_t n_all, uint64_t* first_occ,
                  uint32_t* refcount, void* ws, size_t ws_bytes, void* stream);

/*
 * L3 persistent index (README.md:1288-1292, 1542-1551: "lookup -> found: pointer, refcount++ / new: insert"; sizing
 * README.md:1850-1894).  The table (DEVICE u32[slots], slots = hmse_l3_index_slots(capacity in chunks)) outlives the
 * call: digests [n_old, n_old + n_new) of digests_all join a table that already holds [0, n_old) and are looked up;
 * first_occ[n_old ..) and refcount (running counts, DEVICE u32[>= n_old + n_new]) are updated.  Earlier first
 * occurrences never change, so a batch costs O(n_new) whatever the history.  n_old == 0 clears the table first.
 * Returns HMSE_ENOSPC when the table would exceed load factor 0.5.
 */
uint64_t hmse_l3_index_slots(uint64_t capacity_chunks);
int hmse_l3_index_update(const uint8_t* digests_all, uint64_t n_old, uint64_t n_new, uint64_t* first_occ,
                         uint32_t* refcount, uint32_t* table, uint64_t slots, void* stream);

/*
 * L4a — MinHash signatures.  Replaces minhash_compute(const uint8_t*, size_t,
 * uint32_t*) (README.md:2578-2598): sig[h] = min over 4-byte shingles of
 * MurmurHash3_x86_32(shingle, 4, seed_base + h).
 *   chunk_ids DEVICE u64[n_sel]: chunks to sign (indices into cuts), or NULL = all
 *   sig       DEVICE u32[n_sel][n_hashes]
 *   ws        hmse_workspace_bytes(HMSE_STAGE_L4_MINHASH, n_sel, cfg) bytes hold the call's memo table (per shingle: which
 *             of its 128 hashes are small enough to matter; cleared at the start of every call, so a call depends on
 *             nothing but its arguments).  With a smaller workspace (>= 0 bytes) every hash is computed; the signatures
 *             are the same either way.
 * The table makes the call's TIME data-dependent: it pays where 4-byte shingles repeat across chunks (text: 90 % of the
 * lookups hit, 1.6x faster than without it); on data without repeating shingles (random bytes) the lookups are overhead and
 * a wavefront gives up after 256 of them per pass — measured 6 % slower than a call without workspace.
 * A caller that knows its data is incompressible passes ws_bytes = 0.
 */
int hmse_l4_minhash(const uint8_t* data, uint64_t n, const uint64_t* cuts,
                    const uint64_t* chunk_ids, uint64_t n_sel, const hmse_cfg* cfg,
                    uint32_t* sig, void* ws, size_t ws_bytes, void* stream);

/*
 * L4b — LSH banding (README.md:1375-1383, 1987-1996).  band key = MurmurHash3_x86_32
 * over the band's rows*4 bytes, seed = band index (bucket = key & (2^band_bits - 1));
 * base[i] = earliest j < i sharing a whole band with i, or -1.
 *   band_keys DEVICE u32[n_sel][bands]
 *   base      DEVICE i64[n_sel]  (indices into the selection)
 */
int hmse_l4_lsh(const uint32_t* sig, uint64_t n_sel, const hmse_cfg* cfg, uint32_t* band_keys,
                int64_t* base, void* ws, size_t ws_bytes, void* stream);

/*
 * L4b persistent band tables (README.md:1554-1576 "probe LSH ... none: insert signature", 1937-1945): tables DEVICE
 * u32[bands][slots], slots = hmse_l4_lsh_slots(capacity in stored chunks), outlive the call.  Signatures
 * [n_old, n_old + n_new) of sig_all join tables that already hold [0, n_old); band_keys[n_old ..) are written (or, with
 * keys_given != 0, taken as loaded from a stored band table) and base[n_old ..) = earliest chunk, old or new, sharing a
 * whole band (base may be NULL: insert only).  n_old == 0 clears the tables first.
 */
uint64_t hmse_l4_lsh_slots(uint64_t capacity_chunks);
int hmse_l4_lsh_update(const uint32_t* sig_all, uint64_t n_old, uint64_t n_new, const hmse_cfg* cfg, uint32_t* band_keys,
                       int64_t* base, uint32_t* tables, uint64_t slots, uint32_t keys_given, void* stream);

/*
 * L1 — per-chunk DEFLATE (RFC 1951 raw stream) with the LSH base chunk as preset
 * dictionary.  Replaces mz_deflateInit2(&s, 9, MZ_DEFLATED, 15, 9, ...) +
 * mz_deflate(&s, MZ_FINISH) (README.md:2374, 2378) and the delta rule of
 * README.md:1328, 2175.
 *   chunk_ids DEVICE u64[n_sel] or NULL = all;  base DEVICE i64[n_sel] or NULL
 *             (index into the selection of the dictionary chunk, -1 = none)
 *   out       DEVICE u8[out_cap]; chunk k's stream is out[out_off[k] .. out_off[k+1])
 *   out_off   DEVICE u64[n_sel+1]
 *   kind      DEVICE u8[n_sel]: HMSE_KIND_FULL or HMSE_KIND_DELTA
 *   status    DEVICE u32[1]: bit0 = out_cap overflow (out_off still exact), bit1 = workspace too small
 *   ws        hmse_workspace_bytes(HMSE_STAGE_L1_DEFLATE, n_sel, cfg) is the FIXED part; after it the call needs
 *             one record per chunk: hmse_l1_deflate_record_bytes(len) bytes (about 4*len + 1.4 KiB: histograms and a
 *             token list sized for the all-literal case, which the FULL stream later overwrites), or
 *             hmse_l1_deflate_record_bytes_dict(len) where base >= 0 (len + 21 more: the DELTA stream's own slot)
 */
int hmse_l1_deflate(const uint8_t* data, uint64_t n, const uint64_t* cuts,
                    const uint64_t* chunk_ids, const int64_t* base, uint64_t n_sel,
                    const hmse_cfg* cfg, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                    uint8_t* kind, uint32_t* status, void* ws, size_t ws_bytes, void* stream);

/* Bytes of workspace the record of a chunk of `chunk_len` bytes needs behind the fixed part — without / with a dictionary
 * (0 if chunk_len > 32768: such a chunk cannot be encoded — `uint16_t length`, README.md:1267). */
uint64_t hmse_l1_deflate_record_bytes(uint32_t chunk_len);
uint64_t hmse_l1_deflate_record_bytes_dict(uint32_t chunk_len);

/*
 * L1 with options (the streaming front end): as hmse_l1_deflate, plus
 *   flags  HMSE_DEFLATE_BASE_IS_CHUNK_ID: base[k] is a chunk index into `cuts` (any chunk of the resident data, e.g. one
 *          stored by an earlier batch) instead of an index into the selection.
 */
enum { HMSE_DEFLATE_BASE_IS_CHUNK_ID = 1u };
int hmse_l1_deflate_ex(const uint8_t* data, uint64_t n, const uint64_t* cuts, const uint64_t* chunk_ids,
                       const int64_t* base, uint64_t n_sel, const hmse_cfg* cfg, uint32_t flags, uint8_t* out,
                       uint64_t out_cap, uint64_t* out_off, uint8_t* kind, uint32_t* status, void* ws,
                       size_t ws_bytes, void* stream);

/*
 * Read path, L1 — raw DEFLATE decode of stored chunks.  Replaces mz_inflateInit2(&s, 15) + mz_inflate(&s, MZ_FINISH)
 * (README.md:2397-2400) and the FULL / DELTA branches of the read path (README.md:1635-1669, 2191-2198): a DELTA
 * record inflates with the raw bytes of its base chunk as preset dictionary (the last 32 KiB of them).
 *   streams    DEVICE u8[streams_bytes]
 *   stream_off DEVICE u64[n_sel+1] (dense: stream k = [off[k], off[k+1]))  or, with stream_len != NULL,
 *              DEVICE u64[n_sel] starts + stream_len DEVICE u32[n_sel] (records inside a manifest blob)
 *   kind       DEVICE u8[n_sel]  HMSE_KIND_FULL / HMSE_KIND_DELTA;  base DEVICE i64[n_sel] or NULL: slot of the
 *              dictionary chunk, must be < k (the writer only ever picks earlier chunks)
 *   raw_off    DEVICE u64[n_sel+1]: where each chunk's raw bytes go in raw_out (exclusive prefix sum of raw lengths)
 *   ok         DEVICE u8[n_sel] or NULL: 1 where chunk k decoded, 0 where its record is corrupt
 *   status     DEVICE u32[1]: bit0 = at least one record is corrupt (the streams stock zlib rejects: bad block type or
 *              code set, undefined code, distance beyond the window, stored LEN/NLEN mismatch; plus: decoded size !=
 *              recorded raw length, stream not ending in its last byte, DELTA with a missing/corrupt base) — that
 *              chunk's bytes are undefined, every other chunk is still decoded
 *   ws         hmse_workspace_bytes(HMSE_STAGE_L1_INFLATE, n_sel, cfg)
 */
int hmse_l1_inflate(const uint8_t* streams, uint64_t streams_bytes, const uint64_t* stream_off,
                    const uint32_t* stream_len, const uint8_t* kind, const int64_t* base, uint64_t n_sel,
                    const uint64_t* raw_off, uint8_t* raw_out, uint64_t raw_cap, uint8_t* ok,
                    uint32_t* status, void* ws, size_t ws_bytes, void* stream);

/*
 * Which decoder hmse_l1_inflate launches.  0 (default): by stream count — one stream per wavefront below 49152 streams
 * (a call lasts about as long as its longest stream), one stream per LANE from there on (four times the throughput
 * once the chip's 65 536 lanes are fed).  1 / 2 force the first / second.  Results are identical; process-wide.
 * No reference counterpart: mz_inflate (README.md:2397-2400) decodes one stream per call on one core.
 */
int hmse_l1_inflate_mode(int mode);

/*
 * Read path — POINTER branch and final layout (README.md:1635-1669): chunk i of the original data is the raw
 * bytes of stored slot slot_of_chunk[i] (its own slot for FULL/DELTA, the target's for POINTER).
 *   cuts DEVICE u64[n_chunks+1]; slot_of_chunk DEVICE u64[n_chunks]; raw_off DEVICE u64[n_slots+1]; raw DEVICE u8[]
 *   data_out DEVICE u8[n];  status DEVICE u32[1]: bit0 = a map entry disagrees with the stored lengths
 */
int hmse_read_assemble(const uint64_t* cuts, uint64_t n_chunks, const uint64_t* slot_of_chunk, uint64_t n_slots,
                       const uint64_t* raw_off, const uint8_t* raw, uint8_t* data_out, uint64_t n,
                       uint32_t* status, void* stream);

/*
 * One batch of the streaming front end as a single enqueue with NO host read (BASELINE.json configs[4]
 * "hipGraph-captured per-batch pipeline"; the reference's batch loop README.md:1519-1580: request block -> L2 -> per chunk
 * L3 lookup/insert -> L4 probe -> delta or full).  The batch's bytes are already at data[state[0] .. + batch_bytes); every
 * stage takes its ranges from `state` (DEVICE u64[16]: [0] byte offset, [1] chunks so far, [2] chunks of this batch (out),
 * [3] stored chunks so far, [4] stored chunks of this batch (out), [5] stream bytes so far, [6] stream bytes of this batch
 * (out), [7] sticky status: bit0 chunk capacity, bit1 stored-chunk capacity, bit2 L2 status, bit3 malformed exchange row,
 * bit4 workspace not initialised (hmse_stream_workspace_init), bit5 state block inconsistent (one rank and [8] != [1]: e.g. a stream resumed with the
 * version-1 state layout, whose [8] is 0), bits 8.. DEFLATE status; [8] chunks of ALL ranks so far (== [1] for one rank), [9] chunks of all ranks in this batch (out),
 * [10] global index of this rank's first chunk of the batch (out)), grids and workspace are sized for batch_bytes / min_size
 * chunks, and the call ends by advancing [0], [1], [3], [5], [8] — so the chain can be captured into a hipGraph once per batch
 * size and replayed for every batch.  Once [7] is non-zero the failing batch has been dropped and every later call is a no-op
 * (counters frozen at the last good batch): nothing is ever appended to an index that is no longer consistent.  Appends to the
 * per-chunk arrays of the stream (cuts_all, digests_all, first_occ, refcount, uniq_all, sig_all, band_keys, base_all, kind_all,
 * stream_off_all), to the persistent L3 table / L4 band tables, and writes the batch's DEFLATE streams to out[state[5] ..).
 * seg_off DEVICE u64[n_seg+1]: batch-local segment offsets.  ws: hmse_stream_batch_workspace_bytes(batch_bytes, cfg), prepared ONCE
 * with hmse_stream_workspace_init() and then handed to every batch of the stream unchanged: it holds the MinHash memo table, which
 * persists across the batches (a cache of a pure function of (shingle, cfg->seed_base); cleared per batch it cost every 1 GiB batch
 * its warm-up again).  A chain that finds the workspace untagged drops the batch with status bit4.
 */
uint64_t hmse_stream_batch_workspace_bytes(uint64_t batch_bytes, const hmse_cfg* cfg);
int hmse_stream_workspace_init(void* ws, size_t ws_bytes, uint64_t batch_bytes, const hmse_cfg* cfg, void* stream);
int hmse_stream_batch(uint8_t* data, uint64_t data_cap, uint64_t batch_bytes, const uint64_t* seg_off, uint32_t n_seg,
                      const hmse_cfg* cfg, uint64_t* state, uint64_t* cuts_all, uint64_t max_chunks, uint8_t* digests_all,
                      uint64_t* first_occ, uint32_t* refcount, uint32_t* l3_table, uint64_t l3_slots, uint64_t* uniq_all,
                      uint64_t max_unique, uint32_t* sig_all, uint32_t* band_keys, int64_t* base_all, uint32_t* lsh_tables,
                      uint64_t lsh_slots, uint8_t* kind_all, uint64_t* stream_off_all, uint8_t* out, uint64_t out_cap,
                      void* ws, size_t ws_bytes, void* stream);

/*
 * The same chain for a stream that is sharded over several ranks (one process per GPU; BASELINE.json configs[4] "4 x 10 GB
 * streamed, 8 x MI355X, hipGraph-captured per-batch pipeline").  A global batch is dealt to the ranks as contiguous runs of
 * whole segments ("pieces", rank order = stream order inside the batch), so the GLOBAL CHUNK ORDER is (batch, rank, local
 * index) and equals the natural order of the concatenated stream.  The reference's per-chunk "L3 lookup -> found: pointer /
 * new: insert" (README.md:1538-1551) against ONE index becomes, per batch: phase A on every rank, ONE all-gather of the ranks'
 * exchange rows (RCCL over xGMI; the caller's job — this library never communicates), phase B on every rank.  Every rank
 * keeps a replica of the global digest array and of the L3 table and applies the same order-independent first-occurrence rule
 * to the same rows, so dedupe is that of the one-rank stream; L4 bases and dictionaries are scoped to the rank (the bytes
 * must be resident), exactly as in the sharded one-shot ingest.
 *   cap_bytes    the NOMINAL piece size of the stream (>= every piece_bytes): it alone sizes rows, grids and the workspace, so
 *                that every rank's row has the same layout whatever its piece holds (a rank may get 0 bytes in the last batch)
 *   row          DEVICE u8[hmse_stream_row_bytes(cap_bytes)]: {u64 n_chunks, 24 B zero, n_chunks x 32 B digests, unused tail}
 *   rows         DEVICE u8[world][row bytes]: the all-gathered rows in rank order (world == 1: rows == row)
 *   cuts_all / uniq_all / sig_all / band_keys / base_all / kind_all / stream_off_all / out / lsh_tables: THIS RANK's arrays
 *                (chunk ids are local); gidx DEVICE u64[max local chunks]: global index of every local chunk (out; may be
 *                NULL when world == 1)
 *   digests_g / first_occ_g / refcount_g / l3_table: the GLOBAL arrays (max_chunks_g entries), identical on every rank
 *   state        as hmse_stream_batch: [1]..[6] count this rank's chunks / stored chunks / stream bytes, [8]..[10] the global ones
 *   ws           hmse_stream_batch_workspace_bytes(cap_bytes, cfg); both phases of a batch take the SAME workspace
 * Both calls are stream-ordered and capturable; hmse_stream_batch == hash + encode with world 1 on the row in its workspace.
 */
uint64_t hmse_stream_row_bytes(uint64_t cap_bytes, const hmse_cfg* cfg);
int hmse_stream_piece_hash(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const uint64_t* seg_off,
                           uint32_t n_seg, const hmse_cfg* cfg, uint64_t* state, uint64_t* cuts_all, uint64_t max_chunks,
                           uint8_t* row, void* ws, size_t ws_bytes, void* stream);
int hmse_stream_piece_encode(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const hmse_cfg* cfg,
                             uint64_t* state, const uint8_t* rows, uint32_t world, uint32_t rank, const uint64_t* cuts_all,
                             uint64_t* gidx, uint8_t* digests_g, uint64_t max_chunks_g, uint64_t* first_occ_g,
                             uint32_t* refcount_g, uint32_t* l3_table, uint64_t l3_slots, uint64_t* uniq_all,
                             uint64_t max_unique, uint32_t* sig_all, uint32_t* band_keys, int64_t* base_all,
                             uint32_t* lsh_tables, uint64_t lsh_slots, uint8_t* kind_all, uint64_t* stream_off_all,
                             uint8_t* out, uint64_t out_cap, void* ws, size_t ws_bytes, void* stream);

/*
 * Global L4 for a multi-rank stream as captured phases (round 4; SURVEY.md §8f-3: "cross-GPU base-chunk fetch over xGMI P2P for global L4
 * (config 5)"; the reference keeps ONE set of band tables, README.md:1375-1383).  The stored chunks of all ranks are numbered in global
 * stored order (batch, rank, local); every rank holds the same signature array, band tables and owner map of that numbering and therefore
 * finds, for each of its chunks, the dictionary that ONE rank ingesting the whole stream would find.  Phase B of a batch becomes three
 * enqueue-only calls around two more exchange steps:
 *   hmse_stream_piece_sign      rows of all ranks (as for hmse_stream_piece_encode) -> global index -> this rank's new stored chunks ->
 *                               MinHash -> sig_row {u64 count, u64 first local stored slot, 16 B pad, sig_cap x 512 B}
 *   -- all-gather of the signature rows (fixed size: hmse_stream_sig_row_bytes) --
 *   hmse_stream_piece_bases     sig rows -> global signature array + owner map -> global band tables -> for every new stored chunk of
 *                               this rank its dictionary: a chunk of this rank, or a REMOTE one -> request (owner, owner's stored slot);
 *                               g->req_counts[q] requests to rank q ([world] = total), g->req_slots grouped by owner
 *   -- the requested chunks are fetched (all-to-all; the caller writes the bytes behind its data and their bounds into
 *      cuts_all[g->ghost_chunk0 ..]: request j is chunk ghost_chunk0 + j) --
 *   hmse_stream_piece_encode_g  DEFLATE of the new stored chunks with those dictionaries, tails, both state blocks advanced
 * gstate: DEVICE u64[16], word [3] = stored chunks of ALL ranks before this batch, [4] = of this batch (out), [10] = global stored index
 * of this rank's first new chunk (out).  Status bits as hmse_stream_batch, plus bit7: more new stored chunks than a signature row holds.
 * All arrays are the caller's; sizes in the struct.
 */
typedef struct hmse_gl4 {
  uint32_t struct_size, world, rank, reserved;
  uint64_t sig_cap;        /* hmse_stream_sig_cap(cap_bytes, cfg) */
  uint64_t max_stored_g;   /* capacity of the global stored-chunk arrays */
  uint64_t* gstate;
  uint32_t* sig_g;         /* [max_stored_g][128] */
  uint32_t* band_keys_g;   /* [max_stored_g][bands] */
  int64_t*  base_g;        /* [max_stored_g] global stored index of the dictionary, -1 none */
  uint32_t* lsh_tables_g;  /* [bands][lsh_slots_g], cleared by hmse_l4_lsh_update(n_old = n_new = 0) */
  uint64_t  lsh_slots_g;
  uint32_t* g_owner;       /* [max_stored_g] owning rank */
  uint64_t* g_local;       /* [max_stored_g] the owner's stored slot */
  uint64_t* ug;            /* [max_unique] this rank's stored chunks: global stored index (out, appended) */
  int64_t*  base_global;   /* [max_unique] this rank's stored chunks: base_g (out, appended) */
  uint64_t* req_counts;    /* DEVICE u64[world + 1] (out) */
  uint64_t* req_slots;     /* DEVICE u64[cap chunks of a piece] (out) */
  uint64_t  ghost_chunk0;  /* chunk id of the batch's first fetched dictionary */
} hmse_gl4;
uint64_t hmse_stream_sig_cap(uint64_t cap_bytes, const hmse_cfg* cfg);
uint64_t hmse_stream_sig_row_bytes(uint64_t cap_bytes, const hmse_cfg* cfg);
int hmse_stream_piece_sign(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const hmse_cfg* cfg, uint64_t* state,
                           const uint8_t* rows, uint32_t world, uint32_t rank, const uint64_t* cuts_all, uint64_t* gidx, uint8_t* digests_g,
                           uint64_t max_chunks_g, uint64_t* first_occ_g, uint32_t* refcount_g, uint32_t* l3_table, uint64_t l3_slots,
                           uint64_t* uniq_all, uint64_t max_unique, uint32_t* sig_all, uint8_t* sig_row, void* ws, size_t ws_bytes, void* stream);
int hmse_stream_piece_bases(uint64_t cap_bytes, const hmse_cfg* cfg, uint64_t* state, const uint8_t* sig_rows, const hmse_gl4* g,
                            const uint64_t* uniq_all, uint32_t* band_keys, int64_t* base_all, void* ws, size_t ws_bytes, void* stream);
int hmse_stream_piece_encode_g(uint8_t* data, uint64_t data_cap, uint64_t piece_bytes, uint64_t cap_bytes, const hmse_cfg* cfg, uint64_t* state,
                               uint64_t* gstate, const uint64_t* cuts_all, uint8_t* kind_all, uint64_t* stream_off_all, uint8_t* out,
                               uint64_t out_cap, void* ws, size_t ws_bytes, void* stream);

/*
 * Chunk manifest — the packed on-disk records, written on the GPU (README.md:1263-1270 ChunkIndex 40 B, 2182-2189
 * DeltaChunk 8-byte header + delta data, 1312 pointer 8 B, 1448 per-chunk map, 1635-1669 chunk types).  Replaces the
 * reference's per-chunk "write chunk, insert (sha -> lba, len)" / "pointer record, refcount++" steps of the batch loop
 * (README.md:1542-1551) by one pass over a whole shard; the host only write()s the four arrays.
 *   streams/stream_off/kind/base/uniq_ids   the L1 outputs of the shard's n_unique stored chunks (base: slot index of the dictionary,
 *             -1 none, -2 = a dictionary stored on ANOTHER shard (global L4): the DeltaChunk header is packed unresolved, base_lba
 *             0xFFFFFFFF, and filled in when the shards' manifests are merged; base may be NULL when no record is a DELTA)
 *   digests DEVICE u8[n_chunks][32] or NULL, refcount DEVICE u32[n_chunks] or NULL, cuts DEVICE u64[n_chunks+1]
 *   first_occ DEVICE u64[n_chunks] GLOBAL index of each chunk's first occurrence (NULL: every chunk is its own);
 *             chunk_base = global index of this shard's chunk 0; shard / n_shards (<= 256); shard_bases DEVICE u64[n_shards]
 *             = chunk_base of every shard (NULL when n_shards == 1)
 *   rec_off   DEVICE u64[n_unique+1]: byte offset of each record in the blob (multiples of lba_unit, a power of two; the
 *             record = 8-byte DeltaChunk header for DELTA + the stream); ptr_index DEVICE u64[n_chunks]: number of
 *             POINTER chunks before chunk i
 *   blob      DEVICE u8[blob_bytes]; index DEVICE 40 B x n_unique; chunk_map DEVICE 8 B x n_chunks
 *             {slot u32, raw_length u16, kind u8, shard u8}; pointers DEVICE 8 B x n_pointers
 *             {target_lba u32, target_length u16, flags u16 = HMSE_KIND_POINTER | shard << 4 | 0x8000 if unresolved}.
 *             A chunk whose first occurrence lives on another shard gets slot = that shard's LOCAL chunk index and an
 *             unresolved pointer record (target_lba 0xFFFFFFFF): the merge of the per-shard manifests fills them in.
 *   status    DEVICE u32[1]: bit0 record does not fit (slot, 32-bit lba or 16-bit length), bit1 DELTA without an earlier
 *             base, bit2 first occurrence is not a stored chunk, bit3 forward / unknown cross-shard target, bit4 pointer overflow,
 *             bit5 a stored-chunk id (uniq_ids) outside the shard
 *   ws        hmse_workspace_bytes(HMSE_STAGE_MANIFEST_PACK, n_chunks, cfg)
 */
int hmse_manifest_pack(const uint8_t* streams, const uint64_t* stream_off, const uint8_t* kind, const int64_t* base,
                       const uint64_t* uniq_ids, uint64_t n_unique, const uint8_t* digests, const uint32_t* refcount,
                       const uint64_t* cuts, uint64_t n_chunks, const uint64_t* first_occ, uint64_t chunk_base,
                       uint32_t shard, const uint64_t* shard_bases, uint32_t n_shards, const uint64_t* rec_off,
                       uint32_t lba_unit, const uint64_t* ptr_index, uint8_t* blob, uint64_t blob_bytes, void* index,
                       void* chunk_map, void* pointers, uint64_t n_pointers, uint32_t* status, void* ws, size_t ws_bytes,
                       void* stream);
/* With options: HMSE_MANIFEST_ANY_SHARD_TARGET — a chunk's first occurrence may live on ANY other shard, also a later one.  The
 * shards of a multi-rank stream interleave in stream order (hmse_stream_piece_encode), so the rank that met a chunk first is not
 * always the lower-numbered one; the caller passes first_occ / chunk_base / shard_bases in the STORE numbering (shard, local
 * index) — hmse_amd/stream_dist.py::store_results.  Without the flag a forward target is an error (status bit3), as in a
 * one-shot sharded ingest, where dedupe only ever points backwards. */
enum { HMSE_MANIFEST_ANY_SHARD_TARGET = 1u };
int hmse_manifest_pack_ex(const uint8_t* streams, const uint64_t* stream_off, const uint8_t* kind, const int64_t* base,
                          const uint64_t* uniq_ids, uint64_t n_unique, const uint8_t* digests, const uint32_t* refcount,
                          const uint64_t* cuts, uint64_t n_chunks, const uint64_t* first_occ, uint64_t chunk_base,
                          uint32_t shard, const uint64_t* shard_bases, uint32_t n_shards, uint32_t flags,
                          const uint64_t* rec_off, uint32_t lba_unit, const uint64_t* ptr_index, uint8_t* blob,
                          uint64_t blob_bytes, void* index, void* chunk_map, void* pointers, uint64_t n_pointers,
                          uint32_t* status, void* ws, size_t ws_bytes, void* stream);

/*
 * Diagnostics (bench.py's roofline leg): when enabled, every entry point brackets its DOMINANT
 * kernel launch with a HIP event pair on the caller's stream.  hmse_profile_read() waits for the
 * recorded events (a host sync — never call it inside a capture), adds their durations to the
 * stage's running total and returns it.  Off by default; not part of the data path.
 * Slots: the HMSE_STAGE_* ids (0..31); the six DEFLATE match-kernel size classes report in slots 8..13
 * (S, SG2, SG3, B, S2, SG), their dictionary jobs in 18..23, and the two encode-kernel instantiations in 14 and 15 (FULL
 * records) and 30 and 31 (DELTA records) (hmse_amd/csrc/l1_deflate.hip).
 * hmse_profile_counter(): work counted on the device while profiling is on — the DEFLATE match kernels add the TOKENS they
 * write to their slot (8..13, 18..23), the encode kernels the tokens they read (14, 15, 30, 31):
 * bench.py's algorithmic bytes come from these counts, not from an assumed token density.  A host sync; diagnostics only.
 */
void hmse_profile_enable(int on);
int hmse_profile_read(int stage, double* total_ms, uint64_t* launches, int reset);
int hmse_profile_counter(int slot, uint64_t* value, int reset);

#ifdef __cplusplus
}
#endif
#endif /* HMSE_H */
