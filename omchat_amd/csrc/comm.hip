// Peer all-reduce over IPC-mapped device buffers (xGMI peer-to-peer, one process per GPU) -- the tensor-parallel collective
// for decode-sized messages ("custom peer-store + flag one-shot kernel"), with a two-shot form for large ones.
// New functionality: the reference has no tensor parallelism.  RCCL (model.hip) stays available for every size.
//
// Every rank owns ONE shared allocation (uncached / fine-grained device memory, exported with hipIpcGetMemHandle and mapped by
// every other rank):   [ flags: PEER_MAX_BLOCKS x PEER_MAX_RANKS u32 | data slot 0 | data slot 1 ],  slot = in[cap] + res[cap].
//
// One-shot (<= oneshot_max bytes): block i copies chunk i of the input into its OWN slot (write-through stores), the blocks
// with index i of all ranks meet at a flag barrier, then every rank reads chunk i of EVERY rank's slot and sums in rank order
// 0..n-1 in fp32 (so all ranks produce bit-identical results, and the same bits as a gather-then-sum), one rounding at the end.
// Two-shot: copy-in, barrier, rank r reduces segment r (1/n of the message) from all slots into its result area, barrier,
// every rank gathers the n reduced segments.  Per link and direction that is 2/n of the message, as a ring would move.
//
// Slots alternate with the call parity, which makes ONE barrier per call enough: a rank overwrites slot k&1 in call k+2 only
// after it has left the barrier of call k+1, which every peer enters after its kernel of call k (its reads of that slot) is
// complete (kernels of one rank are stream-ordered).  Barrier = per (block, source rank) monotonic epoch flags; the epoch
// lives in a private per-block counter that the kernel itself advances.
// ORDERING REQUIREMENT: the slot parity is a HOST counter baked into the launch arguments, so all peer kernels of a rank must be issued
// in ONE total order that is the same on every rank, each launched exactly once per host call: never capture a peer call into a
// hipGraph (a replay would repeat the captured parity: two consecutive calls on one slot) and never interleave peer calls from
// unordered streams.  The launchers below refuse a capturing stream; model.hip restricts decode graphs to tp_size == 1 and chains the
// communication stream to the launch stream with events (gemm_allreduce).
// Visibility: payload stores are sc0 sc1 (system-scope write-through) and drained (vmcnt(0)) by every storing wave before the
// workgroup barrier that precedes the flag stores; payload loads are sc0 sc1 (bypass L1 / L2).  Unless the `fast` mode is set,
// one lane additionally issues a system-scope release before the flags and a system-scope acquire after the poll
// (cdna_hip_programming.md Guideline 16).  Every spin is bounded by wall time; a timeout sets a sticky error word that the
// host reads with omchat_peer_error().
#include "kernels.h"
#include "../../include/omchat_hip.h"
#include <string.h>
#include <vector>

namespace {

constexpr int PEER_MAX_RANKS = 8;
constexpr int PEER_MAX_BLOCKS = 128;
constexpr size_t PEER_FLAG_BYTES = 64 * 1024;      // >= PEER_MAX_BLOCKS * PEER_MAX_RANKS * 4, keeps the data 64-KiB aligned
constexpr int PEER_THREADS = 512;
constexpr unsigned long long PEER_TIMEOUT_TICKS = 10ull * 100000000ull;      // 10 s of the 100 MHz wall clock

struct PeerK {
  char* base[PEER_MAX_RANKS];      // every rank's shared allocation as mapped here (base[rank] = own)
  unsigned* ctr;                   // private: per-block barrier epochs
  unsigned* err;                   // private: sticky error word
  int rank, size, fast;
  size_t cap;                      // bytes of one `in` (= one `res`) area
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int AUX_SYS = 17;        // sc0 | sc1: system scope (write-through stores, cache-bypassing loads)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : bytes), 0x00020000);
}

// all workgroups with this block index, one per rank, meet here.  Called by every thread of the block.
__device__ __forceinline__ void peer_barrier(const PeerK& k) {
  __shared__ unsigned s_epoch;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned e = k.ctr[blockIdx.x] + 1u;
    k.ctr[blockIdx.x] = e;
    s_epoch = e;
    if (!k.fast) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const unsigned e = s_epoch;
  if ((int)threadIdx.x < k.size) {
    unsigned* theirs = reinterpret_cast<unsigned*>(k.base[threadIdx.x]) + blockIdx.x * PEER_MAX_RANKS + k.rank;
    __hip_atomic_store(theirs, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned* mine = reinterpret_cast<unsigned*>(k.base[k.rank]) + blockIdx.x * PEER_MAX_RANKS + threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 1023u) == 0u && wall_clock64() - t0 > PEER_TIMEOUT_TICKS) { atomicExch(k.err, 1u); break; }
    }
  }
  if (!k.fast && threadIdx.x == 0) {
    // lane 0 polled its own flag above; the other pollers are in the same wave (size <= 8), so this fence follows every poll
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <typename T> struct Acc8;      // 16 bytes of T <-> fp32 lanes
template <> struct Acc8<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void add(float (&a)[8], u32x4_t v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += __uint_as_float(v[j]);
  }
  static __device__ __forceinline__ u32x4_t pack(const float (&a)[8]) {
    return (u32x4_t){__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])};
  }
};
template <typename T> struct Acc8 {
  static constexpr int N = 8;
  static __device__ __forceinline__ void add(float (&a)[8], u32x4_t v) {
    const typename V8<T>::type x = __builtin_bit_cast(typename V8<T>::type, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += tof(x[j]);
  }
  static __device__ __forceinline__ u32x4_t pack(const float (&a)[8]) {
    typename V8<T>::type x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = fromf<T>(a[j]);
    return __builtin_bit_cast(u32x4_t, x);
  }
};

// sum of the 16-byte pieces at byte offset `off` of area `area_off` of every rank's slot, ranks in ascending order
template <typename T>
__device__ __forceinline__ u32x4_t peer_sum16(const PeerK& k, size_t area_off, int off) {
  u32x4_t v[PEER_MAX_RANKS];
#pragma unroll
  for (int r = 0; r < PEER_MAX_RANKS; ++r)
    if (r < k.size) v[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(k.base[r] + area_off, k.cap), off, 0, AUX_SYS);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < PEER_MAX_RANKS; ++r)
    if (r < k.size) Acc8<T>::add(a, v[r]);
  return Acc8<T>::pack(a);
}

// bytes: message size, multiple of 16.  parity: slot of this call.  Chunks of 16 B are dealt to (block, thread) round-robin.
template <typename T>
__global__ __launch_bounds__(PEER_THREADS) void peer_oneshot_kernel(PeerK k, void* buf, int bytes, int parity) {
  const size_t slot = PEER_FLAG_BYTES + (size_t)parity * 2 * k.cap;
  const __amdgpu_buffer_rsrc_t mine = rsrc_of(k.base[k.rank] + slot, k.cap);
  const int n16 = bytes >> 4;
  const u32x4_t* src = reinterpret_cast<const u32x4_t*>(buf);
  for (int i = blockIdx.x * PEER_THREADS + threadIdx.x; i < n16; i += gridDim.x * PEER_THREADS)
    __builtin_amdgcn_raw_buffer_store_b128(src[i], mine, i << 4, 0, AUX_SYS);
  peer_barrier(k);
  u32x4_t* dst = reinterpret_cast<u32x4_t*>(buf);
  for (int i = blockIdx.x * PEER_THREADS + threadIdx.x; i < n16; i += gridDim.x * PEER_THREADS) dst[i] = peer_sum16<T>(k, slot, i << 4);
}

// two-shot: segment r = 16-B pieces [r * seg16, min((r+1) * seg16, n16))
template <typename T>
__global__ __launch_bounds__(PEER_THREADS) void peer_twoshot_kernel(PeerK k, void* buf, int bytes, int parity) {
  const size_t slot = PEER_FLAG_BYTES + (size_t)parity * 2 * k.cap;
  const __amdgpu_buffer_rsrc_t mine = rsrc_of(k.base[k.rank] + slot, 2 * k.cap);
  const int n16 = bytes >> 4;
  const int seg16 = (n16 + k.size - 1) / k.size;
  const u32x4_t* src = reinterpret_cast<const u32x4_t*>(buf);
  u32x4_t* dst = reinterpret_cast<u32x4_t*>(buf);
  const int stride = gridDim.x * PEER_THREADS, t0 = blockIdx.x * PEER_THREADS + threadIdx.x;
  // Every stage walks a segment as  first piece + t0 + m * stride : piece i of segment r is then touched by the SAME block index on
  // every rank in all three stages (copy-in here, reduction by rank r, gather by the others), which is what the per-block barrier
  // orders.  (A flat copy-in loop over the whole message breaks that whenever a segment is not a whole number of strides: the reducer
  // would read pieces that a DIFFERENT block of the peer has not written yet.)
  for (int r = 0; r < k.size; ++r) {
    const int l1 = r * seg16, h1 = l1 + seg16 < n16 ? l1 + seg16 : n16;
    for (int i = l1 + t0; i < h1; i += stride) __builtin_amdgcn_raw_buffer_store_b128(src[i], mine, i << 4, 0, AUX_SYS);
  }
  peer_barrier(k);
  // reduce my segment into my result area (and into the output)
  const int lo = k.rank * seg16, hi = lo + seg16 < n16 ? lo + seg16 : n16;
  for (int i = lo + t0; i < hi; i += stride) {
    const u32x4_t s = peer_sum16<T>(k, slot, i << 4);
    __builtin_amdgcn_raw_buffer_store_b128(s, mine, (int)k.cap + (i << 4), 0, AUX_SYS);
    dst[i] = s;
  }
  peer_barrier(k);
  // gather the other segments: piece j of segment r was reduced by rank r's thread with the same (block, thread) mapping
  for (int rr = 1; rr < k.size; ++rr) {
    const int r = (k.rank + rr) % k.size;            // start with the next rank: spreads the reads over the links
    const int l2 = r * seg16, h2 = l2 + seg16 < n16 ? l2 + seg16 : n16;
    const __amdgpu_buffer_rsrc_t theirs = rsrc_of(k.base[r] + slot + k.cap, k.cap);
    for (int i = l2 + t0; i < h2; i += stride) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(theirs, i << 4, 0, AUX_SYS);
  }
}

// ---------------------------------------------------------------------------------------------------------
// The two halves of the two-shot all-reduce as collectives of their own (round 6: sequence-parallel norms, model.hip gemm_sp).  The message is
// `size` SEGMENTS -- segment r = seg_bytes at buf + r * seg_stride: the row block rank r owns --
//   reduce-scatter: every rank copies all its segments into its slot, barrier, rank r sums segment r over the slots (rank order, fp32, one rounding:
//                   the bits of the two-shot all-reduce) into ITS segment of buf; the other segments of buf are left as they were;
//   all-gather:     every rank copies its own segment into its slot's result area, barrier, every rank reads the other segments.
// One barrier per call, slots alternate with the call parity exactly as for the all-reduce kernels (same ordering requirement).  Piece i of a
// segment is touched by the same block index on every rank in both stages (first piece + t0 + m * stride), which is what the per-block barrier orders.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PEER_THREADS) void peer_reduce_scatter_kernel(PeerK k, void* buf, int seg_bytes, long seg_stride, int parity) {
  const size_t slot = PEER_FLAG_BYTES + (size_t)parity * 2 * k.cap;
  const __amdgpu_buffer_rsrc_t mine = rsrc_of(k.base[k.rank] + slot, 2 * k.cap);
  const int n16 = seg_bytes >> 4;
  const int stride = gridDim.x * PEER_THREADS, t0 = blockIdx.x * PEER_THREADS + threadIdx.x;
  for (int r = 0; r < k.size; ++r) {
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>((const char*)buf + (size_t)r * seg_stride);
    for (int i = t0; i < n16; i += stride) __builtin_amdgcn_raw_buffer_store_b128(src[i], mine, r * seg_bytes + (i << 4), 0, AUX_SYS);
  }
  peer_barrier(k);
  u32x4_t* dst = reinterpret_cast<u32x4_t*>((char*)buf + (size_t)k.rank * seg_stride);
  for (int i = t0; i < n16; i += stride) dst[i] = peer_sum16<T>(k, slot, k.rank * seg_bytes + (i << 4));
}

__global__ __launch_bounds__(PEER_THREADS) void peer_all_gather_kernel(PeerK k, void* buf, int seg_bytes, long seg_stride, int parity) {
  const size_t slot = PEER_FLAG_BYTES + (size_t)parity * 2 * k.cap;
  const __amdgpu_buffer_rsrc_t mine = rsrc_of(k.base[k.rank] + slot, 2 * k.cap);
  const int n16 = seg_bytes >> 4;
  const int stride = gridDim.x * PEER_THREADS, t0 = blockIdx.x * PEER_THREADS + threadIdx.x;
  const u32x4_t* src = reinterpret_cast<const u32x4_t*>((const char*)buf + (size_t)k.rank * seg_stride);
  for (int i = t0; i < n16; i += stride) __builtin_amdgcn_raw_buffer_store_b128(src[i], mine, i << 4, 0, AUX_SYS);
  peer_barrier(k);
  for (int rr = 1; rr < k.size; ++rr) {
    const int r = (k.rank + rr) % k.size;            // start with the next rank: spreads the reads over the links
    const __amdgpu_buffer_rsrc_t theirs = rsrc_of(k.base[r] + slot, k.cap);
    u32x4_t* dst = reinterpret_cast<u32x4_t*>((char*)buf + (size_t)r * seg_stride);
    for (int i = t0; i < n16; i += stride) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(theirs, i << 4, 0, AUX_SYS);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Tensor-parallel decode: the all-reduce of the split-K slices fused with the residual add + RMSNorm that consumes them.
// Unfused, a row-parallel projection of a decode step is  GEMV (fp32 slices [ks][rows][H]) -> one-shot all-reduce of the slices ->
// resid_rmsnorm_kernel (sums the ks slices, residual, norm): three latency-bound launches.  Here the second and third are one: workgroup
// `row` writes its rank's slices of that row through to its slot, meets the same workgroup of every peer at the flag barrier, then reads
// EVERY rank's slices and finishes the row.  Summation order = the unfused one (per slice s: ranks 0..n-1 from 0.f; then the slices in
// order), so every rank holds the same bits as with omchat_peer_allreduce + launch_resid_rmsnorm (tests/test_gpu_peer.py).
// Slot layout [s][row][H] fp32 (= the GEMV's own layout), one parity per call as for the all-reduce kernels.
// ---------------------------------------------------------------------------------------------------------
constexpr int FN_THREADS = 256;
constexpr int FN_MAXC = 8;            // chunks of 8 elements per thread: H <= 16384

template <typename T>
__global__ __launch_bounds__(FN_THREADS) void peer_resid_rmsnorm_kernel(PeerK k, int parity, T* x, int ldx, const float* part, int ks, int rows,
                                                                      const T* w, T* xn, int ldn, int H, float eps, int pack_nb) {
  typedef typename V8<T>::type v8;
  __shared__ float red[FN_THREADS / 64];
  const int row = blockIdx.x;
  const int nchunk = H >> 3;
  const size_t slot = PEER_FLAG_BYTES + (size_t)parity * 2 * k.cap;
  const __amdgpu_buffer_rsrc_t mine = rsrc_of(k.base[k.rank] + slot, k.cap);
  // independent of the exchange: residual row and norm weights join the first round trip
  v8 xi[FN_MAXC], wv[FN_MAXC];
#pragma unroll
  for (int i = 0; i < FN_MAXC; ++i) {
    const int c = threadIdx.x + i * FN_THREADS;
    if (c < nchunk) {
      xi[i] = ld8<T>(x + (size_t)row * ldx + c * 8);
      if (w) wv[i] = ld8<T>(w + c * 8);
    }
  }
  // my slices of this row -> my slot (write-through)
  for (int s = 0; s < ks; ++s) {
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>(part + ((size_t)s * rows + row) * H);
    const int off = (int)((((size_t)s * rows + row) * H) * 4);
    for (int q = threadIdx.x; q < H / 4; q += FN_THREADS) __builtin_amdgcn_raw_buffer_store_b128(src[q], mine, off + (q << 4), 0, AUX_SYS);
  }
  peer_barrier(k);
  float xv[FN_MAXC][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < FN_MAXC; ++i) {
    const int c = threadIdx.x + i * FN_THREADS;
    if (c < nchunk) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      for (int s = 0; s < ks; ++s) {
        const int off = (int)((((size_t)s * rows + row) * H + c * 8) * 4);
        u32x4_t v0[PEER_MAX_RANKS], v1[PEER_MAX_RANKS];
#pragma unroll
        for (int r = 0; r < PEER_MAX_RANKS; ++r)
          if (r < k.size) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(k.base[r] + slot, k.cap);
            v0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX_SYS);
            v1[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, AUX_SYS);
          }
        float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < PEER_MAX_RANKS; ++r)
          if (r < k.size) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { t[j] += __uint_as_float(v0[r][j]); t[4 + j] += __uint_as_float(v1[r][j]); }
          }
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += t[j];
      }
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rnd<T>(tof(xi[i][j]) + rnd<T>(a[j]));
        xv[i][j] = v; o[j] = fromf<T>(v); ss += v * v;
      }
      st8<T>(x + (size_t)row * ldx + c * 8, o);
    }
  }
  if (!w) return;
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < FN_THREADS / 64; ++q) tot += red[q];
  const float inv = rsqrtf(tot / (float)H + eps);
#pragma unroll
  for (int i = 0; i < FN_MAXC; ++i) {
    const int c = threadIdx.x + i * FN_THREADS;
    if (c < nchunk) {
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[i][j]) * rnd<T>(xv[i][j] * inv));
      st8<T>(pack_nb ? xn + packed_x_index(row, c * 8, pack_nb) : xn + (size_t)row * ldn + c * 8, o);
    }
  }
}

}  // namespace

struct omchat_peer {
  int rank = 0, size = 1, fast = 0;
  size_t cap = 0, total = 0;
  void* local = nullptr;
  void* base[PEER_MAX_RANKS] = {};
  bool opened[PEER_MAX_RANKS] = {};
  unsigned* ctr = nullptr;
  unsigned* err = nullptr;
  unsigned long calls = 0;
  size_t oneshot_max = 256 * 1024;
  int max_blocks = 64;
};

extern "C" int omchat_peer_create(int rank, int size, size_t cap_bytes, omchat_peer** out, char handle_out[64]) {
  OM_CHECK(out && handle_out, "null argument");
  OM_CHECK(size >= 1 && size <= PEER_MAX_RANKS && rank >= 0 && rank < size, "rank / size out of range (<= 8 ranks)");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  static_assert(PEER_MAX_BLOCKS * PEER_MAX_RANKS * 4 <= PEER_FLAG_BYTES, "flag area too small");
  cap_bytes = (cap_bytes + 65535) / 65536 * 65536;
  OM_CHECK(cap_bytes >= 65536 && cap_bytes <= ((size_t)512 << 20), "capacity must be 64 KiB .. 512 MiB (32-bit buffer offsets)");
  omchat_peer* p = new omchat_peer();
  p->rank = rank; p->size = size; p->cap = cap_bytes; p->total = PEER_FLAG_BYTES + 4 * cap_bytes;
  hipError_t e = hipExtMallocWithFlags(&p->local, p->total, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p->local, p->total, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&p->local, p->total); }
  if (e != hipSuccess) { delete p; omchat_set_error(std::string("omchat_peer_create: allocation failed: ") + hipGetErrorString(e)); return 2; }
  auto fail = [&](const char* what, hipError_t er) { omchat_set_error(std::string("omchat_peer_create: ") + what + ": " + hipGetErrorString(er)); (void)hipFree(p->local); delete p; return 2; };
  if ((e = hipMemset(p->local, 0, PEER_FLAG_BYTES)) != hipSuccess) return fail("memset", e);
  if ((e = hipMalloc((void**)&p->ctr, (PEER_MAX_BLOCKS + 16) * 4)) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMemset(p->ctr, 0, (PEER_MAX_BLOCKS + 16) * 4)) != hipSuccess) return fail("memset", e);
  p->err = p->ctr + PEER_MAX_BLOCKS;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return fail("sync", e);
  hipIpcMemHandle_t h;
  memset(&h, 0, sizeof(h));
  if (size > 1) {
    e = hipIpcGetMemHandle(&h, p->local);
    if (e != hipSuccess) { (void)hipGetLastError(); memset(&h, 0, sizeof(h)); }      // same-process groups connect by pointer instead
  }
  memcpy(handle_out, &h, 64);
  p->base[rank] = p->local;
  *out = p;
  return 0;
}

extern "C" int omchat_peer_connect(omchat_peer* p, const char* all_handles) {
  OM_CHECK(p && all_handles, "null argument");
  for (int r = 0; r < p->size; ++r) {
    if (r == p->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, all_handles + (size_t)r * 64, 64);
    void* q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { omchat_set_error(std::string("omchat_peer_connect: hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e)); return 2; }
    p->base[r] = q; p->opened[r] = true;
  }
  return 0;
}

extern "C" void* omchat_peer_base(omchat_peer* p) { return p ? p->local : nullptr; }

extern "C" int omchat_peer_connect_local(omchat_peer* p, void* const* bases) {
  OM_CHECK(p && bases, "null argument");
  for (int r = 0; r < p->size; ++r) if (r != p->rank) { OM_CHECK(bases[r], "null peer base"); p->base[r] = bases[r]; }
  return 0;
}

extern "C" int omchat_peer_set_mode(omchat_peer* p, int fast, size_t oneshot_max_bytes, int max_blocks) {
  OM_CHECK(p, "null argument");
  OM_CHECK(max_blocks >= 0 && max_blocks <= PEER_MAX_BLOCKS, "max_blocks out of range (<= 128)");
  p->fast = fast != 0;
  if (oneshot_max_bytes) p->oneshot_max = oneshot_max_bytes;
  if (max_blocks) p->max_blocks = max_blocks;
  return 0;
}

extern "C" size_t omchat_peer_capacity(omchat_peer* p) { return p ? p->cap : 0; }

static int peer_refuse_capture(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
    omchat_set_error("peer all-reduce launched on a capturing stream: the slot parity is a host counter and is not replay-invariant");
    return 1;
  }
  (void)hipGetLastError();
  return 0;
}

extern "C" int omchat_peer_allreduce(omchat_peer* p, void* buf, size_t count, int dtype, void* stream) {
  OM_CHECK(p && buf, "null argument");
  OM_CHECK(dtype == OMCHAT_F16 || dtype == OMCHAT_BF16 || dtype == 2, "bad dtype");
  if (p->size == 1 || count == 0) return 0;
  for (int r = 0; r < p->size; ++r) OM_CHECK(p->base[r], "peer group is not connected");
  const size_t esz = dtype == 2 ? 4 : 2;
  OM_CHECK(((uintptr_t)buf & 15) == 0 && (count * esz) % 16 == 0, "buffer and byte count must be multiples of 16");
  hipStream_t s = (hipStream_t)stream;
  if (peer_refuse_capture(s)) return 1;
  PeerK k{};
  for (int r = 0; r < p->size; ++r) k.base[r] = (char*)p->base[r];
  k.ctr = p->ctr; k.err = p->err; k.rank = p->rank; k.size = p->size; k.fast = p->fast; k.cap = p->cap;
  size_t done = 0;
  const size_t total = count * esz;
  while (done < total) {
    const size_t piece = total - done < p->cap ? total - done : p->cap;
    const int parity = (int)(p->calls++ & 1);
    char* b = (char*)buf + done;
    const bool one = piece <= p->oneshot_max;
    int grid = (int)((piece / 16 + PEER_THREADS * 4 - 1) / (PEER_THREADS * 4));       // ~4 x 16 B per thread
    grid = grid < 1 ? 1 : (grid > p->max_blocks ? p->max_blocks : grid);
#define OM_PEER_LAUNCH(T)                                                                                                   \
  do {                                                                                                                      \
    if (one) hipLaunchKernelGGL(peer_oneshot_kernel<T>, dim3(grid), dim3(PEER_THREADS), 0, s, k, (void*)b, (int)piece, parity);   \
    else hipLaunchKernelGGL(peer_twoshot_kernel<T>, dim3(grid), dim3(PEER_THREADS), 0, s, k, (void*)b, (int)piece, parity);       \
  } while (0)
    if (dtype == 2) OM_PEER_LAUNCH(float);
    else if (dtype == OMCHAT_F16) OM_PEER_LAUNCH(f16);
    else OM_PEER_LAUNCH(bf16);
#undef OM_PEER_LAUNCH
    OM_LAUNCH_CHECK();
    done += piece;
  }
  return 0;
}

// buf = [size][blk_count] elements.  which = 0: reduce-scatter (block `rank` of buf becomes the sum over the ranks of their block `rank`; the other
// blocks are unchanged), 1: all-gather (every rank's block `rank` is copied into block `rank` of every other rank's buf).  Blocks longer than the
// slot capacity go through in pieces: the same sub-range of every block per launch.
static int peer_rs_ag(omchat_peer* p, void* buf, size_t blk_count, int dtype, int which, void* stream) {
  OM_CHECK(p && buf, "null argument");
  OM_CHECK(dtype == OMCHAT_F16 || dtype == OMCHAT_BF16 || dtype == 2, "bad dtype");
  if (p->size == 1 || blk_count == 0) return 0;
  for (int r = 0; r < p->size; ++r) OM_CHECK(p->base[r], "peer group is not connected");
  const size_t esz = dtype == 2 ? 4 : 2, blk_bytes = blk_count * esz;
  OM_CHECK(((uintptr_t)buf & 15) == 0 && blk_bytes % 16 == 0, "buffer and block byte count must be multiples of 16");
  hipStream_t s = (hipStream_t)stream;
  if (peer_refuse_capture(s)) return 1;
  PeerK k{};
  for (int r = 0; r < p->size; ++r) k.base[r] = (char*)p->base[r];
  k.ctr = p->ctr; k.err = p->err; k.rank = p->rank; k.size = p->size; k.fast = p->fast; k.cap = p->cap;
  const size_t max_piece = (which == 0 ? p->cap / p->size : p->cap) & ~(size_t)15;      // the reduce-scatter parks ALL segments in one slot
  OM_CHECK(max_piece >= 16, "peer slot capacity too small");
  for (size_t done = 0; done < blk_bytes;) {
    const size_t piece = blk_bytes - done < max_piece ? blk_bytes - done : max_piece;
    const int parity = (int)(p->calls++ & 1);
    char* b = (char*)buf + done;
    int grid = (int)((piece / 16 + PEER_THREADS * 4 - 1) / (PEER_THREADS * 4));
    grid = grid < 1 ? 1 : (grid > p->max_blocks ? p->max_blocks : grid);
    if (which == 1) hipLaunchKernelGGL(peer_all_gather_kernel, dim3(grid), dim3(PEER_THREADS), 0, s, k, (void*)b, (int)piece, (long)blk_bytes, parity);
    else if (dtype == 2) hipLaunchKernelGGL(peer_reduce_scatter_kernel<float>, dim3(grid), dim3(PEER_THREADS), 0, s, k, (void*)b, (int)piece, (long)blk_bytes, parity);
    else if (dtype == OMCHAT_F16) hipLaunchKernelGGL(peer_reduce_scatter_kernel<f16>, dim3(grid), dim3(PEER_THREADS), 0, s, k, (void*)b, (int)piece, (long)blk_bytes, parity);
    else hipLaunchKernelGGL(peer_reduce_scatter_kernel<bf16>, dim3(grid), dim3(PEER_THREADS), 0, s, k, (void*)b, (int)piece, (long)blk_bytes, parity);
    OM_LAUNCH_CHECK();
    done += piece;
  }
  return 0;
}
extern "C" int omchat_peer_reduce_scatter(omchat_peer* p, void* buf, size_t blk_count, int dtype, void* stream) { return peer_rs_ag(p, buf, blk_count, dtype, 0, stream); }
extern "C" int omchat_peer_all_gather(omchat_peer* p, void* buf, size_t blk_count, int dtype, void* stream) { return peer_rs_ag(p, buf, blk_count, dtype, 1, stream); }

// x[rows, H] = T(x + T(sum over ranks and slices of part)) in place on every rank, then xn = RMSNorm(x) * w (w == NULL: skip): the fused
// form of  omchat_peer_allreduce(part) + the residual / RMSNorm launch  for split-K slices part = fp32 [ks][rows][H] (ks <= 8)
extern "C" int omchat_peer_resid_rmsnorm(omchat_peer* p, int dtype, void* x, int ldx, const float* part, int ks, const void* w, void* xn, int ldn,
                                         int rows, int H, float eps, int pack_nb, void* stream) {
  OM_CHECK(p && x && part, "null argument");
  OM_CHECK(dtype == OMCHAT_F16 || dtype == OMCHAT_BF16, "bad dtype");
  OM_CHECK(p->size > 1, "single-rank group: use the local kernel");
  for (int r = 0; r < p->size; ++r) OM_CHECK(p->base[r], "peer group is not connected");
  OM_CHECK(rows >= 1 && rows <= PEER_MAX_BLOCKS && ks >= 1 && ks <= 8, "1 <= rows <= 128, 1 <= ks <= 8");
  OM_CHECK(H % 8 == 0 && H <= FN_THREADS * FN_MAXC * 8 && ldx % 8 == 0 && (!w || (xn && ldn % 8 == 0)), "H % 8, H <= 16384, ld % 8");
  OM_CHECK((size_t)ks * rows * H * 4 <= p->cap, "slices exceed the peer slot capacity");
  OM_CHECK(pack_nb == 0 || (rows <= 16 * pack_nb && H % 64 == 0), "packed output: rows <= 16 * NB, H % 64 == 0");
  PeerK k{};
  for (int r = 0; r < p->size; ++r) k.base[r] = (char*)p->base[r];
  k.ctr = p->ctr; k.err = p->err; k.rank = p->rank; k.size = p->size; k.fast = p->fast; k.cap = p->cap;
  hipStream_t s = (hipStream_t)stream;
  if (peer_refuse_capture(s)) return 1;
  const int parity = (int)(p->calls++ & 1);
  if (dtype == OMCHAT_F16)
    hipLaunchKernelGGL(peer_resid_rmsnorm_kernel<f16>, dim3(rows), dim3(FN_THREADS), 0, s, k, parity, (f16*)x, ldx, part, ks, rows, (const f16*)w, (f16*)xn, ldn, H, eps, pack_nb);
  else
    hipLaunchKernelGGL(peer_resid_rmsnorm_kernel<bf16>, dim3(rows), dim3(FN_THREADS), 0, s, k, parity, (bf16*)x, ldx, part, ks, rows, (const bf16*)w, (bf16*)xn, ldn, H, eps, pack_nb);
  OM_LAUNCH_CHECK();
  return 0;
}

// blocks until the device is idle; returns 0 and *err_out = 1 when a barrier spin timed out since the last call (sticky, cleared here)
extern "C" int omchat_peer_error(omchat_peer* p, int* err_out) {
  OM_CHECK(p && err_out, "null argument");
  unsigned e = 0;
  OM_HIP(hipDeviceSynchronize());
  OM_HIP(hipMemcpy(&e, p->err, 4, hipMemcpyDeviceToHost));
  if (e) OM_HIP(hipMemset(p->err, 0, 4));
  *err_out = (int)e;
  return 0;
}

extern "C" void omchat_peer_destroy(omchat_peer* p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < p->size; ++r) if (p->opened[r]) (void)hipIpcCloseMemHandle(p->base[r]);
  if (p->local) (void)hipFree(p->local);
  if (p->ctr) (void)hipFree(p->ctr);
  delete p;
}
