// l3_dedup.hip — L3 exact dedupe over the whole digest array, resident in HBM.
//
// Replaces the linear-probing hash index of README.md:1288-1292 / 1542-1551.
// The reference's found -> refcount++ / new -> insert walk is restated as a pure function of the
// digest array so that it is order-independent (and identical on every rank after the all-gather):
//   first_occ[i] = min{ j : digest[j] == digest[i] },  refcount[first] = multiplicity.
// Open addressing, slot = index of a chunk carrying the slot's digest; a slot is claimed by CAS and
// then only ever lowered with atomicMin by chunks whose 32-byte digest compares equal, so the final
// value is the minimum index whatever the arrival order.
#include "common.h"

constexpr uint32_t DD_EMPTY = 0xFFFFFFFFu;

template <int KEY_U4>  // key size in 16-byte units
__device__ __forceinline__ bool key_equal(const uint8_t* keys, size_t stride, uint32_t a, uint32_t b) {
  const uint8_t* pa = keys + stride * a;
  const uint8_t* pb = keys + stride * b;
  bool eq = true;
#pragma unroll
  for (int i = 0; i < KEY_U4; i++) {
    const uint4 x = load_u4_unaligned(pa + 16 * i), y = load_u4_unaligned(pb + 16 * i);
    eq = eq && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
  }
  return eq;
}

// generic "first occurrence by key content" insert: keys[i] = KEY_U4*16 bytes at keys + stride*i
template <int KEY_U4>
__global__ __launch_bounds__(256) void fo_insert_kernel(const uint8_t* __restrict__ keys, size_t stride,
                                                         const uint32_t* __restrict__ hashes, uint64_t i0, uint64_t n, uint32_t* table,
                                                         uint32_t mask, const uint64_t* __restrict__ st = nullptr) {
  if (st) { i0 = st[0]; n = st[1]; }   // captured chain: {first index, count} read on the device
  const uint64_t i64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i64 >= n) return;
  const uint32_t i = (uint32_t)(i0 + i64);   // entries [i0, i0 + n) join a table that may already hold [0, i0)
  uint32_t hsh;
  if (hashes) hsh = hashes[i];
  else hsh = load_u32_unaligned(keys + stride * i);  // digests are uniform already
  uint32_t slot = hsh & mask;
  for (;;) {
    uint32_t cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == DD_EMPTY) {
      cur = atomicCAS(&table[slot], DD_EMPTY, i);
      if (cur == DD_EMPTY) return;  // claimed
    }
    if (cur == i) return;
    if (key_equal<KEY_U4>(keys, stride, cur, i)) {
      atomicMin(&table[slot], i);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

template <int KEY_U4>
__device__ __forceinline__ uint32_t fo_lookup(const uint8_t* keys, size_t stride, uint32_t hsh, uint32_t i,
                                              const uint32_t* table, uint32_t mask) {
  uint32_t slot = hsh & mask;
  for (;;) {
    const uint32_t cur = table[slot];
    if (cur == DD_EMPTY) return i;  // cannot happen after insert; defensive
    if (cur == i || key_equal<KEY_U4>(keys, stride, cur, i)) return cur;
    slot = (slot + 1) & mask;
  }
}

__global__ __launch_bounds__(256) void dedup_lookup_kernel(const uint8_t* __restrict__ digests, uint64_t i0, uint64_t n,
                                                            const uint32_t* __restrict__ table, uint32_t mask,
                                                            uint64_t* __restrict__ first_occ, uint32_t* __restrict__ refcount,
                                                            const uint64_t* __restrict__ st = nullptr) {
  if (st) { i0 = st[0]; n = st[1]; }   // captured chain: {first index, count} read on the device
  const uint64_t i64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i64 >= n) return;
  const uint32_t i = (uint32_t)(i0 + i64);
  const uint32_t hsh = load_u32_unaligned(digests + 32 * (size_t)i);
  const uint32_t fo = fo_lookup<2>(digests, 32, hsh, i, table, mask);
  first_occ[i] = fo;
  if (refcount) atomicAdd(&refcount[fo], 1u);
}

static uint32_t table_slots(uint64_t n) {
  uint64_t m = 1024;
  while (m < 2 * n) m <<= 1;
  return (uint32_t)m;
}

size_t hmse_l3_dedup_workspace_bytes_impl(uint64_t n) { return hmse_align_up((size_t)table_slots(n) * 4, 256); }

extern "C" int hmse_l3_dedup(const uint8_t* digests_all, uint64_t n_all, uint64_t* first_occ, uint32_t* refcount, void* ws,
                             size_t ws_bytes, void* stream_) {
  if (n_all == 0) return HMSE_OK;
  if (!digests_all || !first_occ) return HMSE_EINVAL;
  if (n_all >= 0x7FFFFFFFull) return HMSE_EINVAL;
  const uint32_t slots = table_slots(n_all);
  if (!ws || ws_bytes < (size_t)slots * 4) return HMSE_ENOSPC;
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();  // drop stale errors of earlier runtime calls made by the host process
  uint32_t* table = (uint32_t*)ws;
  HMSE_FILL(table, 0xFF, (size_t)slots * 4, stream);
  if (refcount) HMSE_FILL(refcount, 0, n_all * sizeof(uint32_t), stream);
  const uint32_t blocks = (uint32_t)((n_all + 255) / 256);
  fo_insert_kernel<2><<<dim3(blocks), dim3(256), 0, stream>>>(digests_all, 32, nullptr, 0, n_all, table, slots - 1);
  HMSE_LAUNCH_CHECK();
  dedup_lookup_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(digests_all, 0, n_all, table, slots - 1, first_occ, refcount);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// Persistent index (README.md:1288-1292 "lookup -> found: refcount++ / new: insert"): the table outlives the
// call.  Digests [n_old, n_old + n_new) are inserted into a table that already holds [0, n_old), then looked up: a new chunk
// equal to an old one finds the old index (smaller indices win the slot), so earlier first occurrences never change and a
// batch costs O(n_new) whatever the history.
extern "C" uint64_t hmse_l3_index_slots(uint64_t capacity_chunks) { return table_slots(capacity_chunks); }

extern "C" int hmse_l3_index_update(const uint8_t* digests_all, uint64_t n_old, uint64_t n_new, uint64_t* first_occ,
                                    uint32_t* refcount, uint32_t* table, uint64_t slots, void* stream_) {
  if (!table || slots < 1024 || (slots & (slots - 1)) || slots > (1ull << 31)) return HMSE_EINVAL;
  if (n_old + n_new >= 0x7FFFFFFFull || 2 * (n_old + n_new) > slots) return HMSE_ENOSPC;   // load factor <= 0.5
  hipStream_t stream = (hipStream_t)stream_;
  (void)hipGetLastError();
  if (n_old == 0) HMSE_FILL(table, 0xFF, (size_t)slots * 4, stream);
  if (n_new == 0) return HMSE_OK;
  if (!digests_all || !first_occ) return HMSE_EINVAL;
  if (refcount) HMSE_FILL(refcount + n_old, 0, n_new * sizeof(uint32_t), stream);
  const uint32_t blocks = (uint32_t)((n_new + 255) / 256);
  fo_insert_kernel<2><<<dim3(blocks), dim3(256), 0, stream>>>(digests_all, 32, nullptr, n_old, n_new, table, (uint32_t)(slots - 1));
  HMSE_LAUNCH_CHECK();
  dedup_lookup_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(digests_all, n_old, n_new, table, (uint32_t)(slots - 1), first_occ, refcount);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

__global__ __launch_bounds__(256) void zero_refcount_kernel(uint32_t* __restrict__ refcount, const uint64_t* __restrict__ st) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < st[1]) refcount[st[0] + i] = 0;
}
// captured chain: range from the device state, grids sized for cap_chunks (the table is never cleared here: the caller's
// first hmse_l3_index_update / a memset did that)
int hmse_l3_index_update_dyn(const uint8_t* digests_all, uint64_t* first_occ, uint32_t* refcount, uint32_t* table, uint64_t slots,
                             const uint64_t* rng, uint64_t cap_chunks, hipStream_t stream) {
  const uint64_t* st = rng;
  if (!digests_all || !first_occ || !refcount || !table || !st || slots < 1024 || (slots & (slots - 1)) || slots > (1ull << 31)) return HMSE_EINVAL;
  const uint32_t blocks = (uint32_t)((cap_chunks + 255) / 256);
  zero_refcount_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(refcount, st);
  fo_insert_kernel<2><<<dim3(blocks), dim3(256), 0, stream>>>(digests_all, 32, nullptr, 0, 0, table, (uint32_t)(slots - 1), st);
  dedup_lookup_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(digests_all, 0, 0, table, (uint32_t)(slots - 1), first_occ, refcount, st);
  HMSE_LAUNCH_CHECK();
  return HMSE_OK;
}

// ---- shared with l4_lsh.hip ----------------------------------------------------------------------
// (explicit instantiation for 128-byte band keys lives there; the template is header-like here)
