// manifest_pack.hip — the chunk manifest's packed records, written on the GPU.
//
// Record layouts are the reference's, little-endian and packed:
//   ChunkIndex  40 B {sha256[32], lba u32, length u16, refcount u16}                       README.md:1263-1270, 2646-2651
//   DeltaChunk   8 B {base_lba u32, base_length u16, delta_length u16} + delta_data[]      README.md:2182-2189
//   pointer      8 B {target_lba u32, target_length u16, flags u16}                         README.md:1312
// plus this build's per-chunk map entry (README.md:1448 "Chunk Map") {slot u32, raw_length u16, kind u8, shard u8}.
// `lba` = byte offset of a record in the shard's blob / lba_unit; every record starts on a multiple of lba_unit.
// A sharded store keeps one blob and one index per shard; a chunk whose first occurrence lives on ANOTHER
// shard becomes a POINTER whose map entry names (shard, that shard's local chunk index) and whose pointer record carries
// the UNRESOLVED flag: the target's lba is only known to its own shard and is filled in when the per-shard manifests are
// merged (hmse_amd/manifest.py merge_manifests) — no second collective in the ingest path.
//
// Two kernels: one workgroup per stored chunk copies its stream into the blob behind the DeltaChunk header and writes its
// ChunkIndex entry; one thread per chunk writes the map entry and, for a duplicate, the pointer record.
#include "common.h"

namespace mfp {

struct Args {
  const uint8_t* streams; const uint64_t* stream_off; const uint8_t* kind; const int64_t* base;
  const uint64_t* uniq_ids; uint64_t n_unique;
  const uint8_t* digests; const uint32_t* refcount;
  const uint64_t* cuts; uint64_t n_chunks;
  const uint64_t* first_occ; uint64_t chunk_base; uint32_t shard;
  const uint64_t* shard_bases; uint32_t n_shards;
  const uint64_t* rec_off; uint32_t lba_unit;
  const uint64_t* ptr_index;
  uint8_t* blob; uint64_t blob_bytes; uint8_t* index; uint8_t* chunk_map; uint8_t* pointers; uint64_t n_pointers;
  uint32_t* slot_of; uint32_t* status;
  uint32_t any_target;   // HMSE_MANIFEST_ANY_SHARD_TARGET: a first occurrence may live on a LATER shard (multi-rank streams)
};

__device__ __forceinline__ uint32_t rec_len_of(const Args& a, uint64_t k) {
  return (uint32_t)(a.stream_off[k + 1] - a.stream_off[k]) + (a.kind[k] == HMSE_KIND_DELTA ? 8u : 0u);
}

__global__ __launch_bounds__(256) void records_kernel(Args a) {
  const uint64_t k = blockIdx.x;
  if (k >= a.n_unique) return;
  const uint32_t t = threadIdx.x;
  const uint64_t s0 = a.stream_off[k], s1 = a.stream_off[k + 1];
  const uint32_t slen = (uint32_t)(s1 - s0);
  const bool delta = a.kind[k] == HMSE_KIND_DELTA;
  const uint64_t o0 = a.rec_off[k], o1 = a.rec_off[k + 1];
  const uint32_t rlen = slen + (delta ? 8u : 0u);
  if (o1 > a.blob_bytes || o0 + rlen > o1 || rlen > 65535u || (o0 % a.lba_unit) != 0 || o0 / a.lba_unit > 0xFFFFFFFFull) {
    if (t == 0) atomicOr(a.status, 1u);   // record does not fit its slot / the 32-bit lba / the 16-bit length
    return;
  }
  uint8_t* dst = a.blob + o0;
  if (delta) {
    if (t == 0) {
      const int64_t b = a.base ? a.base[k] : -1;   // (no base array: every DELTA record is flagged below)
      uint32_t blba = 0xFFFFFFFFu, blen = 0;
      if (b >= 0 && (uint64_t)b < k) { blba = (uint32_t)(a.rec_off[b] / a.lba_unit); blen = rec_len_of(a, (uint64_t)b); }
      else if (b == -2) { }               // dictionary stored on ANOTHER shard (global L4): header left unresolved (lba 0xFFFFFFFF,
                                          // length 0) and filled in by the store merge from that shard's index, like a cross-shard POINTER
      else atomicOr(a.status, 2u);        // a DELTA record needs an EARLIER stored chunk as dictionary
      const uint32_t w0 = blba, w1 = (blen & 0xFFFFu) | (slen << 16);
      __builtin_memcpy(dst, &w0, 4); __builtin_memcpy(dst + 4, &w1, 4);
    }
    dst += 8;
  }
  const uint8_t* src = a.streams + s0;
  for (uint32_t i = t * 16; i < slen; i += 256 * 16) {
    if (i + 16 <= slen) { const uint4 v = load_u4_unaligned(src + i); __builtin_memcpy(dst + i, &v, 16); }
    else for (uint32_t b = i; b < slen; b++) dst[b] = src[b];
  }
  for (uint64_t p = o0 + rlen + t; p < o1; p += 256) a.blob[p] = 0;   // padding up to the next record
  // ChunkIndex entry
  uint8_t* e = a.index + 40 * k;
  const uint64_t c = a.uniq_ids[k];
  if (c >= a.n_chunks) { if (t == 0) atomicOr(a.status, 32u); return; }   // a stored-chunk id outside the shard
  if (t < 32) e[t] = a.digests ? a.digests[32 * c + t] : (uint8_t)0;
  if (t == 32) {
    const uint32_t lba = (uint32_t)(o0 / a.lba_unit);
    uint32_t rc = a.refcount ? a.refcount[c] : 1u;
    if (rc > 65535u) rc = 65535u;
    const uint32_t w = rlen | (rc << 16);
    __builtin_memcpy(e + 32, &lba, 4); __builtin_memcpy(e + 36, &w, 4);
  }
  if (t == 33) a.slot_of[c] = (uint32_t)k;
}

__global__ __launch_bounds__(256) void map_kernel(Args a) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_chunks) return;
  const uint64_t g = a.first_occ ? a.first_occ[i] : a.chunk_base + i;
  uint64_t raw = a.cuts[i + 1] - a.cuts[i];
  if (raw > 65535u) raw = 65535u;
  uint32_t slot, kind, shard = a.shard, ptr_lba = 0xFFFFFFFFu, ptr_len = 0, flags = HMSE_KIND_POINTER;
  bool is_ptr;
  if (g >= a.chunk_base && g < a.chunk_base + a.n_chunks) {       // first occurrence in this shard
    const uint64_t loc = g - a.chunk_base;
    slot = a.slot_of[loc];
    is_ptr = loc != i;
    if (slot == 0xFFFFFFFFu) { atomicOr(a.status, 4u); slot = 0; }   // the first occurrence is not a stored chunk
    kind = is_ptr ? (uint32_t)HMSE_KIND_POINTER : (uint32_t)a.kind[slot];
    if (is_ptr) { ptr_lba = (uint32_t)(a.rec_off[slot] / a.lba_unit); ptr_len = rec_len_of(a, slot); flags |= a.shard << 4; }
  } else {                                                         // ... on another shard: unresolved until the merge
    uint32_t r = 0;
    if (!a.shard_bases || (g >= a.chunk_base && !a.any_target)) { atomicOr(a.status, 8u); slot = 0; r = a.shard; }   // one-shot sharding: dedupe only ever points backwards
    else {
      for (uint32_t q = 1; q < a.n_shards; q++) if (a.shard_bases[q] <= g) r = q;
      slot = (uint32_t)(g - a.shard_bases[r]);
    }
    shard = r; kind = HMSE_KIND_POINTER; is_ptr = true;
    flags |= (r << 4) | 0x8000u;
  }
  uint8_t* m = a.chunk_map + 8 * i;
  const uint32_t w1 = (uint32_t)raw | (kind << 16) | (shard << 24);
  __builtin_memcpy(m, &slot, 4); __builtin_memcpy(m + 4, &w1, 4);
  if (is_ptr) {
    const uint64_t pi = a.ptr_index[i];
    if (pi >= a.n_pointers) { atomicOr(a.status, 16u); return; }
    uint8_t* p = a.pointers + 8 * pi;
    const uint32_t w = (ptr_len & 0xFFFFu) | (flags << 16);
    __builtin_memcpy(p, &ptr_lba, 4); __builtin_memcpy(p + 4, &w, 4);
  }
}

}  // namespace mfp

size_t hmse_manifest_pack_workspace_bytes_impl(uint64_t n_chunks) { return hmse_align_up(4 * (n_chunks + 1), 256); }

extern "C" int hmse_manifest_pack(const uint8_t* streams, const uint64_t* stream_off, const uint8_t* kind, const int64_t* base,
                                  const uint64_t* uniq_ids, uint64_t n_unique, const uint8_t* digests, const uint32_t* refcount,
                                  const uint64_t* cuts, uint64_t n_chunks, const uint64_t* first_occ, uint64_t chunk_base,
                                  uint32_t shard, const uint64_t* shard_bases, uint32_t n_shards, const uint64_t* rec_off,
                                  uint32_t lba_unit, const uint64_t* ptr_index, uint8_t* blob, uint64_t blob_bytes, void* index,
                                  void* chunk_map, void* pointers, uint64_t n_pointers, uint32_t* status, void* ws, size_t ws_bytes,
                                  void* stream_) {
  return hmse_manifest_pack_ex(streams, stream_off, kind, base, uniq_ids, n_unique, digests, refcount, cuts, n_chunks, first_occ, chunk_base, shard,
                               shard_bases, n_shards, 0u, rec_off, lba_unit, ptr_index, blob, blob_bytes, index, chunk_map, pointers, n_pointers, status,
                               ws, ws_bytes, stream_);
}

extern "C" int hmse_manifest_pack_ex(const uint8_t* streams, const uint64_t* stream_off, const uint8_t* kind, const int64_t* base,
                                     const uint64_t* uniq_ids, uint64_t n_unique, const uint8_t* digests, const uint32_t* refcount,
                                     const uint64_t* cuts, uint64_t n_chunks, const uint64_t* first_occ, uint64_t chunk_base,
                                     uint32_t shard, const uint64_t* shard_bases, uint32_t n_shards, uint32_t flags, const uint64_t* rec_off,
                                     uint32_t lba_unit, const uint64_t* ptr_index, uint8_t* blob, uint64_t blob_bytes, void* index,
                                     void* chunk_map, void* pointers, uint64_t n_pointers, uint32_t* status, void* ws, size_t ws_bytes,
                                     void* stream_) {
  using namespace mfp;
  if (flags & ~(uint32_t)HMSE_MANIFEST_ANY_SHARD_TARGET) return HMSE_EINVAL;
  if (!status || lba_unit == 0 || (lba_unit & (lba_unit - 1)) || n_shards == 0 || shard >= n_shards || n_shards > 256) return HMSE_EINVAL;
  if (n_chunks > 0xFFFFFFFEull || n_unique > n_chunks) return HMSE_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  HMSE_FILL(status, 0, sizeof(uint32_t), stream);
  if (n_chunks == 0) return HMSE_OK;
  if (!stream_off || !kind || !uniq_ids || !cuts || !rec_off || !index || !chunk_map || !ptr_index) return HMSE_EINVAL;
  if (n_unique && (!streams || !blob)) return HMSE_EINVAL;
  if (n_pointers && !pointers) return HMSE_EINVAL;
  if (n_shards > 1 && !shard_bases) return HMSE_EINVAL;
  if (!ws || ws_bytes < hmse_manifest_pack_workspace_bytes_impl(n_chunks)) return HMSE_ENOSPC;
  Args a;
  a.streams = streams; a.stream_off = stream_off; a.kind = kind; a.base = base; a.uniq_ids = uniq_ids; a.n_unique = n_unique;
  a.digests = digests; a.refcount = refcount; a.cuts = cuts; a.n_chunks = n_chunks; a.first_occ = first_occ; a.chunk_base = chunk_base;
  a.shard = shard; a.shard_bases = shard_bases; a.n_shards = n_shards; a.rec_off = rec_off; a.lba_unit = lba_unit; a.ptr_index = ptr_index;
  a.blob = blob; a.blob_bytes = blob_bytes; a.index = (uint8_t*)index; a.chunk_map = (uint8_t*)chunk_map; a.pointers = (uint8_t*)pointers;
  a.n_pointers = n_pointers; a.slot_of = (uint32_t*)ws; a.status = status; a.any_target = flags & HMSE_MANIFEST_ANY_SHARD_TARGET;
  HMSE_FILL(a.slot_of, 0xFF, 4 * n_chunks, stream);
  PROF_BEGIN(HMSE_STAGE_MANIFEST_PACK, stream);
  if (n_unique) records_kernel<<<dim3((uint32_t)n_unique), dim3(256), 0, stream>>>(a);
  PROF_END(HMSE_STAGE_MANIFEST_PACK, stream);
  map_kernel<<<dim3((uint32_t)((n_chunks + 255) / 256)), dim3(256), 0, stream>>>(a);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}
