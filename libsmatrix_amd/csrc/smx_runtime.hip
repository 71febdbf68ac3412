// smx_runtime.hip -- host runtime + C ABI of the MI355X-native libsmatrix path.
//
// Owns the HBM tables (directory + row arena), drives the round loop
//   op kernel -> prep -> [row growth] -> [directory growth] -> op kernel on the deferred ops ...
// and exports the reference's eight entry points (include/smatrix.h) plus the
// batched API (include/smatrix_batch.h).  There is NO CPU fallback: without a
// HIP device smatrix_open fails loudly.
#include "smx_kernels.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <linux/falloc.h>
#include <sys/uio.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <condition_variable>
#include <mutex>
#include <string>
#include <map>
#include <vector>

#define SMATRIX_EXPERIMENTAL 1
#include "../../include/smatrix_batch.h"
#include "../../include/smatrix_shard.h"
#include "../../include/smx_probe.h"
#include "smx_stream_priv.h"

using namespace smx;

namespace {

[[noreturn]] void smx_die(const char* msg) {
  // src/smatrix.c:891-894: message on stdout, then abort
  printf("libsmatrix error: %s\n", msg);
  fflush(stdout);
  abort();
}

#define HIP_OK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      char b_[512];                                                                    \
      snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
               __FILE__, __LINE__);                                                    \
      smx_die(b_);                                                                     \
    }                                                                                  \
  } while (0)

inline void chunk_pool_trim();
// hipMalloc that, when the device is out of memory, first gives the pooled chunks of closed matrices back (ChunkPool)
// (SMATRIX_TRACE_ROUNDS: device allocations and frees of this thread, counted and timed -- a first batch makes dozens)
struct AllocClock { uint64_t n_alloc = 0, n_free = 0; double s_alloc = 0, s_free = 0; };
inline AllocClock& alloc_clock() { static thread_local AllocClock c; return c; }
inline double mono_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
inline bool trace_alloc() { static const bool on = getenv("SMATRIX_TRACE_ROUNDS") && *getenv("SMATRIX_TRACE_ROUNDS") == '3'; return on; }   // (SMATRIX_TRACE_ROUNDS=3: ... and a line per device allocation / free)
inline hipError_t dev_free(void* p) {
  const double t0 = mono_s();
  const hipError_t e = hipFree(p);
  AllocClock& c = alloc_clock(); c.n_free++; c.s_free += mono_s() - t0;
  if (trace_alloc()) fprintf(stderr, "[smatrix]     free %p  %.3f ms\n", p, (mono_s() - t0) * 1e3);
  return e;
}
// The library's OWN stream-ordered pool, one per device (DevBuf::need_on: scratch of one batch).  Freed memory stays in it up
// to 1 GB instead of going back to the driver at the next synchronisation.  (Round 4 raised the release threshold of the
// device's DEFAULT pool instead -- a process-wide side effect on the host application)  nullptr when the runtime
// refuses to create one: need_on then falls back to plain allocations.
inline hipMemPool_t scratch_pool() {
  static std::mutex mu;
  static std::vector<std::pair<int, hipMemPool_t>> pools;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> g(mu);
  for (auto& e : pools) if (e.first == dev) return e.second;
  hipMemPoolProps props = {};
  props.allocType = hipMemAllocationTypePinned;
  props.handleTypes = hipMemHandleTypeNone;
  props.location.type = hipMemLocationTypeDevice;
  props.location.id = dev;
  hipMemPool_t pool = nullptr;
  if (hipMemPoolCreate(&pool, &props) != hipSuccess) { (void)hipGetLastError(); pool = nullptr; }
  if (pool) {
    uint64_t keep = 1ull << 30;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    (void)hipGetLastError();
  }
  pools.push_back({dev, pool});
  return pool;
}
// what the pool retains goes back to the driver (before the library gives up on an allocation)
inline void scratch_pool_trim() {
  if (hipMemPool_t pool = scratch_pool()) { (void)hipMemPoolTrimTo(pool, 0); (void)hipGetLastError(); }
}

// Buffers that DevBuf::need has outgrown are not handed back on the spot: hipFree waits for the device and for the driver
// (1.7 ms for the ten buffers the dense stream's third batch replaces -- in the middle of its rounds).  They wait here, at most
// GRAVEYARD_MAX bytes of them (need() doubles, so they add up to less than what replaced them), until a matrix is closed, an
// allocation fails, or smatrix_release_cached_memory is called.
struct Graveyard {
  static constexpr size_t GRAVEYARD_MAX = (size_t)8 << 30;
  std::mutex mu;
  std::vector<std::pair<void*, size_t>> dead;
  size_t bytes = 0;
  void bury(void* p, size_t n) {
    std::unique_lock<std::mutex> g(mu);
    dead.push_back({p, n});
    bytes += n;
    if (bytes > GRAVEYARD_MAX) { g.unlock(); empty(); }
  }
  void empty() {
    std::vector<std::pair<void*, size_t>> d;
    { std::lock_guard<std::mutex> g(mu); d.swap(dead); bytes = 0; }
    for (auto& e : d) (void)dev_free(e.first);
  }
};
inline Graveyard& graveyard() { static Graveyard g; return g; }

template <typename T>
inline void dev_malloc(T** p, size_t bytes) {
  struct Tick { double t0 = mono_s(); size_t b; ~Tick() { AllocClock& c = alloc_clock(); c.n_alloc++; c.s_alloc += mono_s() - t0;
                  if (trace_alloc()) fprintf(stderr, "[smatrix]     alloc %zu bytes  %.3f ms\n", b, (mono_s() - t0) * 1e3); } } tick;
  tick.b = bytes;
  if (hipMalloc(reinterpret_cast<void**>(p), bytes) == hipSuccess) return;
  (void)hipGetLastError();
  graveyard().empty();
  chunk_pool_trim();
  scratch_pool_trim();
  HIP_OK(hipMalloc(reinterpret_cast<void**>(p), bytes));
}

inline uint32_t blocks_for(uint64_t n, uint32_t per = 256) {
  return (uint32_t)std::max<uint64_t>(1, (n + per - 1) / per);
}

// ---- row arena: one contiguous VA range, physical memory mapped on demand ------
// zero-fill in pieces of at most 1 GiB (a single > 4 GiB fill faulted when the arena grew by one, see Arena)
inline void zero_async(void* p, size_t bytes, hipStream_t st) {
  for (size_t off = 0; off < bytes; off += (size_t)1 << 30)
    HIP_OK(hipMemsetAsync(static_cast<uint8_t*>(p) + off, 0, std::min<size_t>((size_t)1 << 30, bytes - off), st));
}

// Physical chunks of closed matrices are kept for the next one of the same process (up to SMATRIX_CHUNK_POOL_GB, default 64;
// 0 = give everything back at once).  Memory handed back to the driver is wiped before it can be handed out again, and an
// allocation that lands on memory still waiting for that wipe blocks: a 12.5 GB growth step of a matrix opened right after
// a 27 GB one had been closed was measured at 3.2 s against the usual 2 ms.  Chunks come in a few fixed sizes so that they fit
// again; the arena zero-fills whatever it maps, reused or fresh.
struct ChunkPool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, hipMemGenericAllocationHandle_t> free_chunks;    // (device, bytes) -> handle
  size_t bytes = 0, cap = (size_t)64 << 30;
  bool cap_read = false;
  bool take(int dev, size_t n, hipMemGenericAllocationHandle_t* h) {
    std::lock_guard<std::mutex> g(mu);
    auto it = free_chunks.find({dev, n});
    if (it == free_chunks.end()) return false;
    *h = it->second;
    free_chunks.erase(it);
    bytes -= n;
    return true;
  }
  bool put(int dev, size_t n, hipMemGenericAllocationHandle_t h) {
    std::lock_guard<std::mutex> g(mu);
    if (!cap_read) {
      if (const char* e = getenv("SMATRIX_CHUNK_POOL_GB")) cap = (size_t)strtoull(e, nullptr, 10) << 30;
      cap_read = true;
    }
    if (bytes + n > cap) return false;
    free_chunks.insert({{dev, n}, h});
    bytes += n;
    return true;
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& c : free_chunks) (void)hipMemRelease(c.second);
    free_chunks.clear();
    bytes = 0;
  }
};
inline ChunkPool& chunk_pool() { static ChunkPool* p = new ChunkPool; return *p; }     // (never destroyed: no HIP calls at exit)
inline void chunk_pool_trim() { chunk_pool().trim(); }

struct Arena {
  uint8_t* base = nullptr;
  size_t mapped = 0;       // bytes usable
  size_t reserved = 0;     // VA reserved (vmm) -- 0 in fallback mode
  size_t gran = 0;
  bool vmm = false;
  int device = 0;
  std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> chunks;

  void init(int dev, size_t first_bytes, hipStream_t st) {
    device = dev;
    const char* no = getenv("SMATRIX_NO_VMM");
    if (!(no && *no == '1')) try_vmm();
    grow_to(first_bytes, 0, st);
  }

  void try_vmm() {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) {
      (void)hipGetLastError();
      return;
    }
    g = std::max<size_t>(g, 2u << 20);   // keep chunks 2 MiB aligned (the probe reports 4 KiB)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    // a row base is a 32-bit count of 128-byte units: 512 GiB addressable
    size_t want = std::min<size_t>(total_b, (size_t)UNIT_BYTES << 32);
    want = (want + g - 1) / g * g;
    void* p = nullptr;
    if (hipMemAddressReserve(&p, want, g, nullptr, 0) != hipSuccess || !p) {
      (void)hipGetLastError();
      return;
    }
    base = static_cast<uint8_t*>(p);
    reserved = want;
    gran = g;
    vmm = true;
  }

  // make [0, bytes) usable; `live` bytes must be preserved (fallback mode copies them)
  void grow_to(size_t bytes, size_t live, hipStream_t st) {
    if (bytes <= mapped) return;
    if (vmm) {
      HIP_OK(hipStreamSynchronize(st));   // nothing may be running on the range while its access is re-set
      size_t total = (bytes - mapped + gran - 1) / gran * gran;
      if (mapped + total > reserved) smx_die("row arena exhausted (all of HBM reserved)");
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned;
      prop.location.type = hipMemLocationTypeDevice;
      prop.location.id = device;
      hipMemAccessDesc ad = {};
      ad.location.type = hipMemLocationTypeDevice;
      ad.location.id = device;
      ad.flags = hipMemAccessFlagsProtReadWrite;
      // physical chunks (and the fills that zero them) of at most 1 GiB each: a growth step of a 10+ GB
      // arena is several GiB, and single > 4 GiB allocations / fills are not something to depend on
      // (in a few fixed sizes -- 1 GiB, 256, 64, 16, 4 MiB, then single granules -- so that the chunks of a closed matrix fit
      //  the next one's growth steps: ChunkPool)
      while (total) {
        size_t add = gran;
        for (size_t sz = (size_t)1 << 30; sz >= ((size_t)4 << 20); sz >>= 2)
          if (sz % gran == 0 && sz <= total) { add = sz; break; }
        hipMemGenericAllocationHandle_t h;
        if (!chunk_pool().take(device, add, &h) && hipMemCreate(&h, add, &prop, 0) != hipSuccess) {
          // out of memory while chunks of closed matrices sit idle in the pool (they only fit requests of their own
          // size): hand them back to the driver and try once more before dying
          (void)hipGetLastError();
          chunk_pool().trim();
          HIP_OK(hipMemCreate(&h, add, &prop, 0));
        }
        HIP_OK(hipMemMap(base + mapped, add, 0, h, 0));
        // Access for the new chunk only where the runtime takes a sub-range (the HIP 7.0 runtime bundled
        // with PyTorch does; there the whole-range form costs ~8 ms per mapped GB: 30-60 ms per growth of a
        // 5-7 GB arena).  ROCm 7.2's own runtime rejects a sub-range with hipErrorInvalidValue
        // (tools/probe/vmm.cpp) and does the whole range in microseconds -- so: try, then fall back.
        if (hipMemSetAccess(base + mapped, add, &ad, 1) != hipSuccess) {
          (void)hipGetLastError();
          HIP_OK(hipMemSetAccess(base, mapped + add, &ad, 1));
        }
        HIP_OK(hipMemsetAsync(base + mapped, 0, add, st));
        chunks.push_back({h, add});
        mapped += add;
        total -= add;
      }
    } else {
      size_t nb = std::max(bytes, mapped * 2);
      uint8_t* p = nullptr;
      dev_malloc(&p, nb);
      zero_async(p, nb, st);
      if (base && live) HIP_OK(hipMemcpyAsync(p, base, live, hipMemcpyDeviceToDevice, st));
      HIP_OK(hipStreamSynchronize(st));
      if (base) HIP_OK(hipFree(base));
      base = p;
      mapped = nb;
    }
  }

  void destroy() {
    if (vmm) {
      size_t off = 0;
      for (auto& c : chunks) {
        (void)hipMemUnmap(base + off, c.second);
        if (!chunk_pool().put(device, c.second, c.first)) (void)hipMemRelease(c.first);
        off += c.second;
      }
      if (base) (void)hipMemAddressFree(base, reserved);
    } else if (base) {
      (void)hipFree(base);
    }
    base = nullptr;
    mapped = 0;
    chunks.clear();
  }
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  bool pooled = false;                 // allocated with hipMallocAsync (need_on)
  void need(size_t n) {
    if (n <= cap) return;
    if (p) graveyard().bury(p, cap * sizeof(T));      // (not freed here: see Graveyard)
    size_t c = std::max<size_t>(n, cap * 2);
    dev_malloc(&p, c * sizeof(T));
    cap = c;
    pooled = false;
  }
  void release() {
    if (p) (void)dev_free(p);          // (hipFree also takes stream-ordered allocations: it waits for the device)
    p = nullptr;
    cap = 0;
    pooled = false;
  }
  // Stream-ordered variants for scratch that lives for one batch and is used on ONE stream (the cold start's key set and
  // lists): hipFree of >= 64 MB costs 0.21 ms whatever the device is doing (tools/probe/free_cost.cpp) -- seven of them were
  // 1.5 ms of a 9.6 ms first batch -- while hipFreeAsync hands the memory back to the device's pool in 5 us.
  void need_on(size_t n, hipStream_t s) {
    if (n <= cap) return;
    release_on(s);
    size_t c = n;
    const double t0 = mono_s();
    static const bool use_pool = !(getenv("SMATRIX_SCRATCH_POOL") && *getenv("SMATRIX_SCRATCH_POOL") == '0');
    hipMemPool_t pool = use_pool ? scratch_pool() : nullptr;
    if (pool && hipMallocFromPoolAsync(reinterpret_cast<void**>(&p), c * sizeof(T), pool, s) == hipSuccess) {
      pooled = true;
      AllocClock& ac = alloc_clock(); ac.n_alloc++; ac.s_alloc += mono_s() - t0;
      if (trace_alloc()) fprintf(stderr, "[smatrix]     alloc (stream-ordered) %zu bytes  %.3f ms\n", c * sizeof(T), (mono_s() - t0) * 1e3);
    } else {
      (void)hipGetLastError();
      p = nullptr;
      dev_malloc(&p, c * sizeof(T));
      pooled = false;
    }
    cap = c;
  }
  void release_on(hipStream_t s) {
    if (p && pooled) {
      const double t0 = mono_s();
      HIP_OK(hipFreeAsync(p, s));
      AllocClock& ac = alloc_clock(); ac.n_free++; ac.s_free += mono_s() - t0;
      if (trace_alloc()) fprintf(stderr, "[smatrix]     free (stream-ordered) %p  %.3f ms\n", (void*)p, (mono_s() - t0) * 1e3);
    } else if (p) {
      (void)dev_free(p);
    }
    p = nullptr;
    cap = 0;
    pooled = false;
  }
};

constexpr size_t SCALAR_ROW_PAIRS_ALLOC = 8192 + 8;       // pinned pair buffer of the scalar getrow (+ slack for S4's overrun)

struct ScalarReq {
  int op;
  uint32_t x, y, v, result;
  bool done;
};

// ---- host-side write-back cell cache of the scalar ABI --------------------------------------------------
// The reference's callers issue ONE op per call (JNI, Ruby, src/smatrix_benchmark.c:29-65) and mostly return to
// the same cells; a device round trip per call is ~15-20 us, ~360x the reference on its own benchmark.  Cells
// the device is KNOWN to hold (a scalar call created or read them) are therefore mirrored here: get/set/incr/decr
// on such a cell is host arithmetic on the mirrored value (exactly the reference's: :230/:241/:252, wrapping
// uint32) and only marks the entry dirty; dirty values go back in one batched set before anything else can look
// at the tables (batch calls, getrow, row dumps, close).
//  * only value words are deferred -- a cell that is not known to exist takes the device path AT ONCE, so the
//    order of insertions, and with it every table's byte layout, stays that of the call sequence;
//  * y == 0 never enters (quirk Q1-Q3 cells change meaning with their value), and a y == 0 WRITE drops the whole
//    cache: it can turn a (0,v) cell into an empty one and cut probe chains (the reference then no longer finds
//    keys behind the cut, and neither may a cached entry);
//  * batch writes drop it as well (they may touch mirrored cells).
struct CellCache {
  // Round 3: calls on mirrored cells take NO lock.  JVM threads hammer a few hundred shared cells
  // (src/smatrix_benchmark.c:29-65: overlapping id blocks); behind one mutex per shard eight threads spent their time
  // in futex waits (1 M gets: 95-190 ms against the reference's 27 ms, whose readers only bump a counter).  Now:
  //  * an entry is {key, word} of two 64-bit atomics in a table that is allocated once at full size and never moves;
  //    word = value | (generation << 2 | state) << 32 with state 0 empty / 1 clean / 2 dirty;
  //  * get = one load of the word; set / incr / decr = ONE compare-and-swap of the word (new value, state dirty), so
  //    every caller gets the value after ITS op (:230/:241/:252, wrapping), exactly, under any interleaving;
  //  * inserts and wipes are the rare, locked operations (shard mutex; their callers hold the matrix lock);
  //  * a drain takes an entry's dirty value with an exchange-like CAS of the whole word -- dirty -> clean when the
  //    mirror is kept, -> empty when it is dropped -- so a write either lands before (and is in the drained value) or
  //    finds the word changed and starts over / takes the slow path: nothing is lost, nobody waits;
  //  * the generation (bumped by every wipe of the shard) keeps a write that was aimed at a forgotten entry from
  //    landing on whatever was mirrored in that slot afterwards.
  static constexpr unsigned SHARDS = 16;
  static constexpr size_t SLOTS = (size_t)1 << 18;     // per shard, fixed (4 MB of zero pages, touched as they fill)
  struct Ent {
    std::atomic<uint64_t> key;                          // x << 32 | y
    std::atomic<uint64_t> word;                         // value | (generation << 2 | state) << 32
  };
  static uint32_t w_val(uint64_t w) { return (uint32_t)w; }
  static uint32_t w_state(uint64_t w) { return (uint32_t)(w >> 32) & 3u; }
  static uint64_t w_make(uint32_t val, uint32_t gen, uint32_t state) { return (uint64_t)val | ((uint64_t)((gen << 2) | state) << 32); }
  struct alignas(64) Shard {
    std::atomic<uint32_t> dirty{0};                     // entries that went clean -> dirty since the last drain (a hint)
    uint32_t gen = 1;                                   // bumped by every wipe (30 bits are kept in the words)
    std::mutex mu;                                      // insert / wipe
    Ent* tab = nullptr;                                 // SLOTS entries, calloc'ed at the first insert
    std::vector<uint32_t> occ;                          // occupied slots (what a drain walks)
  };
  struct alignas(64) Stripe { std::atomic<uint64_t> n{0}; };
  Shard sh[SHARDS];
  Stripe hit_stripes[64];                               // hit counts, striped by thread: the counter must not become the shared line
  bool enabled = true;
  size_t shard_cap = (size_t)1 << 17;                   // entries per shard before the shard is recycled (2 M cells in all)
  std::atomic<uint64_t> flushes{0}, flushed_cells{0};
  ~CellCache() { for (Shard& s : sh) free(s.tab); }

  uint64_t hits_total() const { uint64_t t = 0; for (const Stripe& s : hit_stripes) t += s.n.load(std::memory_order_relaxed); return t; }
  void count_hit() {
    static std::atomic<uint32_t> next{0};
    thread_local uint32_t mine = next.fetch_add(1, std::memory_order_relaxed) & 63u;
    hit_stripes[mine].n.fetch_add(1, std::memory_order_relaxed);
  }
  static uint64_t mix(uint32_t x, uint32_t y) {
    uint64_t z = ((uint64_t)x << 32 | y) * 0x9e3779b97f4a7c15ULL;
    z ^= z >> 29; z *= 0xbf58476d1ce4e5b9ULL; z ^= z >> 32;
    return z;
  }
  // the entry of `key`, or nullptr; *w = its word as seen (state != 0)
  static Ent* find(Shard& s, uint64_t h, uint64_t key, uint64_t* w) {
    Ent* tab = s.tab;
    if (!tab) return nullptr;
    for (size_t i = (h >> 4) & (SLOTS - 1);;) {
      Ent& e = tab[i];
      const uint64_t word = e.word.load();
      if (w_state(word) == 0) return nullptr;
      if (e.key.load(std::memory_order_acquire) == key) {
        // Validated like a seqlock: the word is read AGAIN behind the key.  A slot's key changes only after a
        // wipe (word -> 0, then key, then a word of the NEXT generation), so a key read between two loads that show the
        // same generation, both non-empty, belongs to those words.  Without the second load a reader that stalled between
        // the two loads could pair the NEW key of a recycled slot with the OLD entry's value -- and a get would return
        // another cell's value (writes were safe: their CAS fails on the generation).
        const uint64_t again = e.word.load();
        if (again == word || (w_state(again) != 0 && (again >> 34) == (word >> 34))) { *w = again; return &e; }
        continue;                                        // recycled under us: look at this slot again
      }
      i = (i + 1) & (SLOTS - 1);
    }
  }
  // the host arithmetic of one op on a mirrored cell; false = not mirrored
  bool apply(int op, uint32_t x, uint32_t y, uint32_t v, uint32_t* res) {
    if (!enabled || y == 0) return false;
    const uint64_t h = mix(x, y), key = (uint64_t)x << 32 | y;
    Shard& s = sh[h & (SHARDS - 1)];
    uint64_t w;
    Ent* e = find(s, h, key, &w);
    if (!e) return false;
    if (op == OP_GET) {
      if (e->key.load(std::memory_order_relaxed) != key) return false;          // (recycled between the two loads of find)
      // (a frozen entry -- state 3, put() is recycling the shard -- still holds the cell's value)
      *res = w_val(w);
      count_hit();
      return true;
    }
    for (;;) {
      if (w_state(w) == 3) return false;                                        // frozen: the shard is being recycled
      const uint32_t nv = op == OP_SET ? v : op == OP_INCR ? w_val(w) + v : w_val(w) - v;
      const uint64_t nw = (w & 0xFFFFFFFC00000000ull) | ((uint64_t)2 << 32) | nv;   // same generation, dirty, new value
      if (e->word.compare_exchange_weak(w, nw)) {
        if (w_state(w) == 1) s.dirty.fetch_add(1, std::memory_order_relaxed);
        *res = nv;
        count_hit();
        return true;
      }
      // the word changed under us: another writer (go again on its value), a drain that kept the entry (state 1 now:
      // go again), or a wipe (state 0 / another generation: this key is no longer mirrored here)
      if (w_state(w) == 0 || ((w ^ nw) >> 34) != 0 || e->key.load(std::memory_order_relaxed) != key) return false;
    }
  }
  void wipe_locked(Shard& s) {                          // caller holds s.mu
    for (uint32_t i : s.occ) s.tab[i].word.store(0);
    s.occ.clear();
    s.dirty.store(0, std::memory_order_relaxed);
    s.gen = (s.gen + 1) & 0x3FFFFFFFu;
    if (s.gen == 0) s.gen = 1;
  }
  // the device holds `val` in cell (x,y): mirror it (clean).  Caller holds the matrix lock.
  void put(uint32_t x, uint32_t y, uint32_t val) {
    if (!enabled || y == 0) return;
    const uint64_t h = mix(x, y), key = (uint64_t)x << 32 | y;
    Shard& s = sh[h & (SHARDS - 1)];
    std::lock_guard<std::mutex> g(s.mu);
    if (!s.tab) {
      s.tab = static_cast<Ent*>(calloc(SLOTS, sizeof(Ent)));          // all-zero = all empty
      if (!s.tab) smx_die("malloc() failed");
    }
    uint64_t w;
    if (Ent* e = find(s, h, key, &w)) {                                 // (a dirty entry is newer than the device)
      if (w_state(w) == 1) e->word.compare_exchange_strong(w, (w & 0xFFFFFFFF00000000ull) | val);   // (lost to a writer: its value is newer)
      return;
    }
    if (s.occ.size() >= shard_cap) {
      // full: recycle the shard, all or nothing.  Every clean entry is FROZEN first (state 3: readers still use it, writers
      // take the slow path -- which waits for the matrix lock our caller holds); one dirty entry and everything is thawed
      // again (its value has to reach the device first: until the next drain has been through).  Only a completely
      // frozen shard is wiped: a half-wiped one would hide entries behind cut probe sequences while they are still live.
      size_t frozen = 0;
      bool dirty_found = false;
      for (; frozen < s.occ.size() && !dirty_found; frozen++) {
        Ent& e = s.tab[s.occ[frozen]];
        uint64_t w = e.word.load();
        for (;;) {
          if (w_state(w) == 2) { dirty_found = true; break; }
          if (e.word.compare_exchange_weak(w, w | ((uint64_t)3 << 32))) break;      // 1 -> 3
        }
        if (dirty_found) break;
      }
      if (dirty_found) {
        for (size_t k = 0; k < frozen; k++) {
          Ent& e = s.tab[s.occ[k]];
          e.word.store(e.word.load() & ~((uint64_t)2 << 32));                         // 3 -> 1 (nobody else writes a frozen word)
        }
        return;
      }
      wipe_locked(s);
    }
    size_t i = (h >> 4) & (SLOTS - 1);
    while (w_state(s.tab[i].word.load(std::memory_order_relaxed))) i = (i + 1) & (SLOTS - 1);
    s.tab[i].key.store(key, std::memory_order_relaxed);
    s.tab[i].word.store(w_make(val, s.gen, 1));                         // published
    s.occ.push_back((uint32_t)i);
  }
  // dirty cells -> out (marked clean); clear: forget everything afterwards.  Caller holds the matrix lock.
  void drain(std::vector<uint32_t>& xs, std::vector<uint32_t>& ys, std::vector<uint32_t>& vs, bool clear) {
    for (Shard& s : sh) {
      if (!s.tab || s.occ.empty()) continue;
      std::lock_guard<std::mutex> g(s.mu);
      if (!clear && !s.dirty.load(std::memory_order_acquire)) continue;
      s.dirty.store(0, std::memory_order_release);
      for (uint32_t i : s.occ) {
        Ent& e = s.tab[i];
        uint64_t w = e.word.load();
        for (;;) {
          if (!clear && w_state(w) != 2) break;
          const uint64_t nw = clear ? 0ull : (w & 0xFFFFFFFC00000000ull) | ((uint64_t)1 << 32) | w_val(w);
          if (e.word.compare_exchange_weak(w, nw)) {                    // took exactly the value that was there
            if (w_state(w) == 2) {
              const uint64_t k = e.key.load(std::memory_order_relaxed);
              xs.push_back((uint32_t)(k >> 32)); ys.push_back((uint32_t)k); vs.push_back(w_val(w));
            }
            break;
          }
        }
      }
      if (clear) {
        s.occ.clear();
        s.gen = (s.gen + 1) & 0x3FFFFFFFu;
        if (s.gen == 0) s.gen = 1;
      }
    }
  }
};

struct Matrix {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::mutex qmu;                       // scalar-call combining (see scalar_op)
  std::condition_variable qcv;
  std::vector<ScalarReq*> queue;
  bool combining = false;

  Ctl* d_ctl = nullptr;
  Ctl* h_ctl = nullptr;        // pinned
  DirSlot* d_dir = nullptr;
  uint32_t dir_size = 0;
  Arena arena;
  uint64_t arena_next = 1;     // host mirror (unit 0 is reserved: base 0 = "no block")
  uint32_t dir_used = 0;       // host mirror

  DevBuf<uint32_t> defer[2];
  DevBuf<GrowTask> tasks;
  uint64_t io_window = 256ull << 20;    // bytes per window of the file loader / writer
  uint32_t in_stride = 1;               // words between consecutive ops of the batch being applied (3 / 2: packed records)
  uint32_t prep_blocks = 512;           // grid cap of k_prep (it loops): most launches have far fewer deferred ops than the batch had ops
  unsigned io_threads = 16;             // host threads of the file loader / writer (SMATRIX_IO_THREADS; 1 = the serial code)
  DevBuf<uint32_t> klist;               // growth tasks of kind k at [k * klist_cap, ...), k = 0..2
  uint32_t klist_cap = 0;
  DevBuf<uint32_t> rebal;
  FreeLists fl = {};                    // their pointers/capacities as passed to the kernels
  int32_t free_cnt[N_CLASSES] = {0};    // host mirror of the stack heights (last readback)
  DevBuf<uint32_t> map_old, map_new;
  DevBuf<unsigned long long> disp_mask; // per old chunk of a growth round: its displaced cells (k_grow_move_home -> k_grow_rest_lds)
  DevBuf<uint32_t> rest_tab;            // k_grow_rest_plan: {slices, then per slice: task, slice | slices of the row << 16}
  uint32_t rest_slice_cells = REST_SLICE_CELLS;     // SMATRIX_REST_SLICE (tests: slices of 64 cells, one slice per row)
  static constexpr uint32_t rest_grid = 768;        // (swept 256..1536 with slices of 256..2048 cells: 6.26-6.52 ms per dense-id step, flat)
  DevBuf<uint64_t> cellp;
  DevBuf<uint32_t> sx, sy, sv, so;      // staging for the host-pointer API
  DevBuf<uint64_t> soff;
  bool no_ret = false;                  // the write batch in flight has no result array (d_out == NULL, CF import)
  uint32_t set_entries = 0;             // set batch in flight: entries of k_set_fold (0: the batch went op by op)
  bool set_always_locate = false;       // SMATRIX_SET_LOCATE=1 (tests, A/B): the entry passes always start with k_set_locate_e
  DevBuf<uint32_t> ent_idx;
  DevBuf<uint32_t> big;                 // getrow: rows too large for one wave
  DevBuf<uint32_t> seg;                 // getrow: their segments (first segment per row, then a count per segment)
  uint32_t* d_small = nullptr;          // 16 words of scratch
  uint32_t* h_small = nullptr;          // pinned

  smatrix_stats_t st = {};
  uint32_t agg_min = 1024;              // batches at least this long fold duplicates in LDS first
  // retry rounds hold mostly distinct new keys and are small: the one-op-per-lane kernel spreads them
  // over many more workgroups (measured: +4.5 % on config 2 with the threshold at 3e5..2e6)
  uint32_t agg_min_retry = 1u << 20;
  // ... but only for lists of keys that wait for a row to DOUBLE.  A list the bulk path handed back, or one whose rows did
  // not exist a round ago, holds the batch's own ops, duplicates and all: 587 000 incr(5, 0, 1) of a 2^20-op batch went lane
  // by lane through the column-0 compare-and-swap loop (quirk Q1) on ONE cell -- minutes instead of milliseconds (round 4:
  // found when the bulk path's hand-back fell just below 2^20 ops; round 3's batches sat at the threshold by luck)
  bool retry_may_repeat = false;
  bool profile = false;
  bool trace_rounds = false;            // SMATRIX_TRACE_ROUNDS=1: one stderr line per round
  bool trace_sync = false;              // SMATRIX_TRACE_ROUNDS=2: ... and a stream sync + a line after every launch (DBG_STEP)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  struct TimedLaunch { hipEvent_t e0, e1; uint32_t n; };
  std::deque<TimedLaunch> get_pending;  // profiled get launches whose events have not been read yet (get_timing_resolve)
  std::vector<hipEvent_t> ev_free;
  static constexpr bool grow_fork = true;   // (the chunked passes of a growth round on a helper stream beside the in-LDS rehashes: see grow_rows)
  hipStream_t helper = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;

  std::string fname;
  // file mode: something changed since the file was loaded / last written.  Set by every writer (the scalar mirror's
  // host-side writes included, which hold no lock); whoever flushes takes it with exchange(false) BEFORE it syncs the
  // mirror and collects the dirty rows, so a write that lands during a flush keeps the flag up for the next one.
  std::atomic<bool> dirty{false};
  bool in_cache_sync = false;           // the write batch in flight is the mirror's write-back (its cells were flagged when they were written)
  // background flusher (the reference's IO thread, src/smatrix.c:929-960: dirty rows reach the file behind the caller's
  // back, 100 ms poll :945): a timer thread that runs cache_sync + file_flush under the matrix lock every
  // SMATRIX_FLUSH_MS (default 100; 0 = off), never more than ~1/10 of the time (the pause grows with the last flush)
  uint64_t flush_ms = 100;
  // flushes that release the matrix lock while they write (file_flush, snapshot mode): the FILE and its index belong to the
  // holder of file_mu.  Lock order (round 6): file_mu FIRST, then m->mu -- whoever wants to flush queues for the
  // file without the matrix lock, so a write in flight (up to 2 GB under file_mu alone) never has a thread waiting for it
  // with m->mu in its hand and every caller of the handle behind that thread.  Nobody takes file_mu while holding m->mu.
  std::mutex file_mu;
  std::atomic<bool> ckpt_due{false};         // SMATRIX_FLUSH_EVERY: the call that is finishing owes a checkpoint (CkptAfter: taken once m->mu is released)
  uint64_t flush_snapshot = 2048ull << 20;   // bytes of row tables one such flush snapshots on the device (SMATRIX_FLUSH_SNAPSHOT_MB)
  hipStream_t flush_stream = nullptr;        // its copies to the host
  std::atomic<uint64_t> file_flushes_done{0}, file_rows_written_done{0}, file_bg_flushes_done{0};
  std::thread flusher;
  std::mutex fl_mu;
  std::condition_variable fl_cv;
  bool fl_stop = false;
  CellCache cache;                      // scalar ABI: mirrored cells (see CellCache)
  DevBuf<uint64_t> row_ret;             // scalar getrow: pooled device buffer for rows that outgrow the pinned one
  uint32_t* h_row = nullptr;            // pinned: {count, spare, big[2], offsets[2] (u64)} + pairs written by the kernel itself
  void* file_index = nullptr;           // FileIndex (smx_file.inc): where every row lives in the backing file
  bool compact_at_close = false;        // SMATRIX_COMPACT_AT_CLOSE=1: close rewrites the file without leaked blocks
  bool file_fsync = false;              // SMATRIX_FSYNC=1: fsync between the row blocks and the CMAP entries, and after
  uint64_t flush_every = 0;             // SMATRIX_FLUSH_EVERY=N: checkpoint the file after every N write batches
  uint64_t dbg_after = 0;               // SMATRIX_DBG_AFTER: batch number from which a measurement build's debug mode applies
  // the bulk path (k_fix_*): taken in round 0 when the previous write batch deferred a large share of its ops
  bool bulk_enabled = true;             // SMATRIX_BULK=0 switches it off
  bool expect_bulk = true;              // an empty matrix creates its rows: expect it
  uint32_t fix_share = 4;               // ... and only when at least 1/fix_share of the batch is pending (SMATRIX_BULK_SHARE)
  uint32_t fix_presize_min = 1u << 18;  // deferred ops from which the rows to create are counted first (SMATRIX_BULK_PRESIZE_MIN; ~0: never)
  uint32_t fix_min = 1u << 14;          // deferred ops from which the grouping pays (SMATRIX_BULK_MIN): its fixed cost is ~6 small
                                        // launches and 3 read-backs, about two rounds of the loop it replaces
  DevBuf<uint32_t> fx_cnt, fx_cur, fx_pos, fx_touched, fx_where, fx_grouped, fx_rank;
  std::vector<void*> put_aside;         // device blocks replaced while kernels were running: freed at the end of the write batch
  uint32_t fx_dir_size = 0;             // directory size fx_cnt / fx_cur / fx_pos were laid out (and zeroed) for
  DevBuf<uint64_t> fx_excl, fx_tiles;
  // the speculative chain (run_write): on when the previous write batch was finished by its round 1
  bool spec_enabled = true;             // SMATRIX_SPEC=0 switches it off
  bool spec_ready = false;
  bool spec_tiny = false;               // SMATRIX_SPEC_TINY=1 (tests)
  uint32_t last_nd0 = 1;                // ops the previous write batch deferred in its round 0 (0: the next batch is not chained)
  uint32_t spec_nd_prev = 0, spec_nt_prev = 0, spec_nk_prev[4] = {0, 0, 0, 0};   // the previous batch's round 0: deferred ops, growth tasks (by kind)
  uint64_t spec_gu_prev = 0;            // ... and the units its growths took
  bool long_probes = false;             // this batch: the folding kernel set ops aside for the wave-cooperative probe -> retries run lane-per-op
  // cold starts (insert_pending_keys): a large deferred list is reduced to its distinct keys once, and the rounds that the
  // hot rows' doublings need run over those
  // clustered rows: set for good once a batch has shown long probe sequences (dense ids); SMATRIX_CLUSTERED=1 / 0 forces it
  bool clustered = false, clustered_forced = false;
  uint32_t clustered_quiet = 0;         // chained batches in a row whose (sampled) count of long probes stayed below 1/256 of the batch
  // where far-from-home keys sit (smx_kernels.hpp ArenaHead): 2^hint_lg slots of 16 bytes = two {tag, cell} entries each, allocated
  // when the tables turn out clustered (256 MB; 64 MB of one-entry slots until round 6: the hot rows' cold keys lost their hints);
  // SMATRIX_HINT_LG (0: no hints)
  uint4* d_hints = nullptr;
  uint32_t hint_lg = 24;
  static constexpr uint32_t wpo_max = 1u << 22;   // retry lists up to this length run a wave per op on clustered tables
  uint32_t* absent_list_dev = nullptr;  // mirror of ArenaHead::absent_list
  static constexpr bool retry_far = true;   // the long retry lists of a clustered table in host-driven rounds go through a far join of their own
  static constexpr bool small_first = true; // the cold rounds of a clustered table take the keys below their row's size first (list order: 88 vs 72.6 ms)
  static constexpr bool retry_split = true; // the retry of a clustered table in two halves: k_apply_short, then a wave per op over the rest (0.78 -> 0.66 ms)
  static constexpr bool absent_split = true;   // the clustered folding kernel keeps two deferred lists (8.5 -> 7.9 ms)
  unsigned long long* rest_dbg = nullptr; uint32_t rest_dbg_mode = 0; uint64_t rest_dbg_from = 0;   // SMATRIX_REST_DBG (measurement runs: k_grow_rest_lds)
  static constexpr bool get_split = true;   // gets of a clustered table ask for the next cells and the hint in one trip (k_get_clu: 0.90 -> 0.80 ms)
  bool rest_lds = true;                 // SMATRIX_REST_LDS=0: clustered rows' displaced cells move by priority probing alone (k_grow_move_rest), as in round 4
  void* host_pipe = nullptr;            // HostPipe: the staging of large host-pointer batches (smatrix_apply_batch and friends)
  // the far join of a clustered write batch (smx_kernels.hpp "far join"): SMATRIX_FAR_JOIN=0 switches it off
  bool far_join = true;
  DevBuf<uint4> far_tab;
  DevBuf<uint32_t> far_unit_row, far_zeros;
  DevBuf<unsigned long long> far_occ, far_occ0;
  DevBuf<uint32_t> far_clm;
  // (round 6) growth takes the waiting keys in (growth.hpp, k_pend_group): prep's records, the keys grouped by growing row, the
  // directory slot -> task map, the set that keeps one op per key, {records, bucket bump pointer}
  DevBuf<uint2> pend_rec;
  DevBuf<uint32_t> pend_keys, task_of, pend_ctl;
  DevBuf<unsigned long long> pend_hash;
  double w_call_s = 0, w_wait_s = 0, w_alloc_s = 0;      // write batches: wall time inside run_write, of it waiting for the device, of it in device allocations / frees
  bool pend_on = true;                  // SMATRIX_PEND=0: the keys that wait for a doubling go in through the retry, as in round 5
  bool pend_armed = false;              // the prep that has just been enqueued left records (the growth round that follows groups them)
  uint32_t pend_est = 0;                // ... about so many
  DevBuf<uint32_t> big_list;            // [0] = n, then the directory slots of the rows of >= 2^FAR_ROW_LG cells (k_far_rows rebuilds it, k_grow_commit appends)
  bool big_list_valid = false;          // ... complete for the directory as it stands (not after a rebuild of the directory or a file load)
  DevBuf<uint2> far_unit_info;          // per unit: its row's block, size and its place in the row (k_far_rows -> k_far_scan)
  DevBuf<uint32_t> far_bloom;           // one bit per key of F (k_far_keys -> k_far_scan)
  DevBuf<uint32_t> far_rcnt, far_bucket, far_prows;   // k_far_absent / k_far_place: absent keys per row (at its first unit; + the row count), their entries of F, the rows that have any
  bool far_place = true;                // SMATRIX_FAR_PLACE=0: the keys the join calls absent are inserted one by one by the pass (claims by rank), as in round 5
  uint32_t far_tab_lg = 0;              // what ArenaHead's far fields name
  uint32_t far_nd_seen = 0;             // ops in the list the last join was built for
  uint32_t far_rows_seen = 0, far_units_seen = 0;   // rows of >= 2^HOME_LG cells / their 1024-cell units when they were last counted
  bool home_on = false;                 // the rows' at-home bitmaps (smx_kernels.hpp HOME_LG) are kept up to date and used by the probes: host mirror of ArenaHead::home_on
  uint32_t cold_min = 1u << 20;         // deferred ops from which it is tried (SMATRIX_COLD_MIN; 0 = never)
  uint32_t cold_share = 64;             // ... and only when at least 1/cold_share of the batch is still pending (SMATRIX_COLD_SHARE).  (Not stricter: the
                                        // first chunk of the CF import has 3 M of 2^25 ops pending when its rows have been created, and needs it:
                                        // 0.036 against 0.126 s.  The price is 0.12 ms of de-duplication that finds nothing in batch 2 of config 2.)
  DevBuf<unsigned long long> cold_set;
  DevBuf<uint32_t> cold_idx[2], cold_zero;   // the walks of a cold round through the far join: the list 0..n-1, what the pass leaves, zeroed amounts
  static constexpr uint32_t cold_far_max = 1u << 20;   // ... in rounds of up to this many keys: the early rounds' millions of keys sit in small rows, a lane walks them faster
  static constexpr bool cold_all_far = true;   // every walker of an indexed row enters the join's table (only the long probes: 29 vs 22 ms)
  static constexpr bool cold_far = true;    // the walks of a clustered table's cold rounds go through the far join (a wave per key: 37 vs 69 ms)
  DevBuf<unsigned long long> cold_keys[3];   // the distinct pending keys, packed; what a round leaves deferred
};

void set_device(Matrix* m) { HIP_OK(hipSetDevice(m->device)); }

// SMATRIX_TRACE_ROUNDS=2: wait for the stream after every launch of a write batch and say which one it was -- the last line
// before a GPU fault names the kernel
#define DBG_STEP(m, s, what)                                                          \
  do {                                                                                \
    if ((m)->trace_sync) {                                                            \
      fprintf(stderr, "[smatrix]     %s ...", what); fflush(stderr);                  \
      HIP_OK(hipStreamSynchronize(s));                                                \
      if ((m)->helper) HIP_OK(hipStreamSynchronize((m)->helper));                     \
      fprintf(stderr, " done\n");                                                    \
    }                                                                                 \
  } while (0)

void ctl_reset_round(Matrix* m, hipStream_t s) {
  HIP_OK(hipMemsetAsync(m->d_ctl, 0, CTL_ROUND_BYTES, s));     // the persistent part stays on the device
}

// host -> device for the persistent part (open, file load)
void ctl_push_persistent(Matrix* m, hipStream_t s) {
  Ctl c = {};
  c.dir_used = m->dir_used;
  c.arena_next = m->arena_next;
  memcpy(c.free_cnt, m->free_cnt, sizeof c.free_cnt);
  *m->h_ctl = c;
  HIP_OK(hipMemcpyAsync(m->d_ctl, m->h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, s));
  HIP_OK(hipStreamSynchronize(s));
}

// the host waits for the device inside a write batch (a read-back): counted, so that a caller can tell the time its batch spent
// with kernels running from the time the host took between them (smatrix_stats_t::write_wait_ms)
void sync_counted(Matrix* m, hipStream_t s) {
  const double t0 = mono_s();
  HIP_OK(hipStreamSynchronize(s));
  m->w_wait_s += mono_s() - t0;
}
void ctl_read(Matrix* m, hipStream_t s) {
  HIP_OK(hipMemcpyAsync(m->h_ctl, m->d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, s));
  sync_counted(m, s);
  m->dir_used = m->h_ctl->dir_used;
  m->arena_next = m->h_ctl->arena_next;
  memcpy(m->free_cnt, m->h_ctl->free_cnt, sizeof m->free_cnt);
}

// room for `extra` more entries on the stack of class c (contents preserved)
void ensure_free_cap(Matrix* m, uint32_t c, uint64_t extra, hipStream_t s) {
  const uint64_t have = m->free_cnt[c] > 0 ? (uint64_t)m->free_cnt[c] : 0;
  const uint64_t need = have + extra;
  if (need <= m->fl.cap[c]) return;
  // (more than what is asked for: a young table's task counts creep up from batch to batch -- 116 818, 118 736, ... -- and a stack
  //  sized exactly was reallocated in every one of them, each time with a hipFree that waits for the device: 0.4 ms a batch)
  // (a quarter on top, not twice -- every size class gets room for ALL tasks of a round, 7 stacks of 2 x 4 bytes per
  //  task were 0.9 GB at 2^24 rows; the creep is a few % per batch)
  const uint64_t ncap = std::max<uint64_t>(need + std::max<uint64_t>(need / 4, 65536), (uint64_t)m->fl.cap[c] * 5 / 4);
  uint32_t* np = nullptr;
  dev_malloc(&np, ncap * sizeof(uint32_t));
  if (have) HIP_OK(hipMemcpyAsync(np, m->fl.list[c], have * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  HIP_OK(hipStreamSynchronize(s));
  if (m->fl.list[c]) HIP_OK(dev_free(m->fl.list[c]));
  m->fl.list[c] = np;
  m->fl.cap[c] = (uint32_t)std::min<uint64_t>(ncap, 0x7fffffffu);
}

void ensure_arena_free(Matrix* m, uint64_t units, hipStream_t s) {
  uint64_t need = (m->arena_next + units) * UNIT_BYTES;
  if (need <= m->arena.mapped) return;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  struct Report {
    Matrix* m; struct timespec* t0; struct timespec* t1; hipStream_t s;
    ~Report() {
      if (!m->trace_rounds) return;
      (void)hipStreamSynchronize(s);
      clock_gettime(CLOCK_MONOTONIC, t1);
      fprintf(stderr, "[smatrix]   arena grown to %.2f GB in %.3f ms\n", m->arena.mapped / 1e9,
              (t1->tv_sec - t0->tv_sec) * 1e3 + (t1->tv_nsec - t0->tv_nsec) * 1e-6);
    }
  } report{m, &t0, &t1, s};
  // grow geometrically so that mapping calls stay rare
  uint64_t target = std::max<uint64_t>(need, m->arena.mapped + m->arena.mapped / 2);
  if (m->arena.vmm) target = std::min<uint64_t>(std::max<uint64_t>(need, target), m->arena.reserved);
  m->arena.grow_to(target, m->arena_next * UNIT_BYTES, s);
}

constexpr uint32_t BIG_LIST_CAP = 1u << 20;
void grow_directory(Matrix* m, uint32_t factor, hipStream_t s) {
  m->big_list_valid = false;                                // (directory slots move)
  uint64_t ns64 = (uint64_t)m->dir_size * factor;
  if (ns64 > (1ull << 31)) smx_die("directory too large");
  uint32_t ns = (uint32_t)ns64;
  DirSlot* nd = nullptr;
  dev_malloc(&nd, (size_t)ns * sizeof(DirSlot));
  zero_async(nd, (size_t)ns * sizeof(DirSlot), s);
  hipLaunchKernelGGL(k_dir_rehash, dim3(blocks_for(m->dir_size)), dim3(256), 0, s, m->d_dir,
                     m->dir_size, nd, ns - 1);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(dev_free(m->d_dir));
  m->d_dir = nd;
  m->dir_size = ns;
  m->st.dir_grown++;
}

// the hint table of a clustered matrix (ArenaHead); the words in unit 0 of the arena are what the kernels read
void ensure_hints(Matrix* m, hipStream_t s) {
  if (m->d_hints || m->hint_lg == 0) return;
  const size_t bytes = (size_t)16 << m->hint_lg;
  dev_malloc(&m->d_hints, bytes);
  zero_async(m->d_hints, bytes, s);
  struct { uint32_t mask; uint4* p; } __attribute__((packed)) w = {(1u << m->hint_lg) - 1u, m->d_hints};
  static_assert(sizeof(w) == 12 && offsetof(ArenaHead, hint_mask) == 4 && offsetof(ArenaHead, hints) == 8, "ArenaHead layout");
  HIP_OK(hipMemcpyAsync(m->arena.base + offsetof(ArenaHead, hint_mask), &w, sizeof(w), hipMemcpyHostToDevice, s));
  HIP_OK(hipStreamSynchronize(s));                         // (`w` is on the stack)
  if (m->trace_rounds) fprintf(stderr, "[smatrix] clustered tables: hint table of 2^%u entries\n", m->hint_lg);
}

// a 32-bit word of ArenaHead (unit 0 of the arena)
void arena_head_set(Matrix* m, size_t offset, uint32_t value, hipStream_t s) {
  HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m->arena.base + offset), (int)value, 1, s));
}
// ArenaHead::absent_list: where the clustered folding kernel puts the ops that wait for prep (nullptr: one deferred list)
void absent_list_set(Matrix* m, uint32_t* list, hipStream_t s) {
  if (m->absent_list_dev == list) return;
  m->absent_list_dev = list;
  const uint64_t v = reinterpret_cast<uint64_t>(list);
  arena_head_set(m, offsetof(ArenaHead, absent_list), (uint32_t)v, s);
  arena_head_set(m, offsetof(ArenaHead, absent_list) + 4, (uint32_t)(v >> 32), s);
}

// The device-side helpers of a clustered matrix brought in line with m->clustered: the hint table, and the rows' at-home
// bitmaps (smx_kernels.hpp HOME_LG) -- nobody sets bits while a matrix is not clustered, so they are rebuilt from the tables
// as they stand at the moment it turns out to be (the tables must be quiescent: between two rounds, or after a load).
void clustered_sync(Matrix* m, hipStream_t s) {
  if (m->clustered) {
    ensure_hints(m, s);
    if (m->home_on) return;
    if (m->dir_used) {
      DevBuf<uint32_t> list;
      list.need((size_t)m->dir_used + 1);
      HIP_OK(hipMemsetAsync(m->d_small + 14, 0, 4, s));
      hipLaunchKernelGGL(k_home_list, dim3(std::min<uint32_t>(blocks_for(m->dir_size), 4096)), dim3(256), 0, s, m->d_dir, m->dir_size, list.p, m->d_small + 14,
                         m->dir_used);
      HIP_OK(hipGetLastError());
      HIP_OK(hipMemcpyAsync(m->h_small + 14, m->d_small + 14, 4, hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      const uint32_t n = std::min(m->h_small[14], m->dir_used);
      for (uint32_t first = 0; first < n; first += 32768) {
        hipLaunchKernelGGL(k_home_rebuild, dim3(64, std::min<uint32_t>(n - first, 32768)), dim3(256), 0, s, m->d_dir, list.p, first, m->arena.base);
        HIP_OK(hipGetLastError());
      }
      HIP_OK(hipStreamSynchronize(s));
      list.release();
      if (m->trace_rounds) fprintf(stderr, "[smatrix] clustered tables: at-home bitmaps of %u rows rebuilt\n", n);
    }
    arena_head_set(m, offsetof(ArenaHead, home_on), 1u, s);
    m->home_on = true;
  } else if (m->home_on) {
    arena_head_set(m, offsetof(ArenaHead, home_on), 0u, s);
    m->home_on = false;
  }
}

// The far join in front of the wave-per-op pass of a clustered write batch (smx_kernels.hpp "far join"): the table and the
// occupancy bitmap are (re)built on stream s for the deferred list `dl` (length on the device: ctl->n_prev); every capacity is an
// estimate -- what does not fit is left out of the table and takes the old walk.  Returns false when nothing was enqueued.
bool far_join_enqueue(Matrix* m, hipStream_t s, const uint32_t* dl, const uint32_t* x, const uint32_t* y, uint32_t est_nd, bool all_far = false) {
  if (!m->far_join || !m->home_on || m->dir_used == 0) return false;
  if (m->far_units_seen == 0) {
    // the first join of this matrix: how many rows and units there are is counted once, with a read-back
    HIP_OK(hipMemsetAsync(&m->d_ctl->n_big, 0, 8, s));
    hipLaunchKernelGGL(k_far_rows, dim3(std::min<uint32_t>(blocks_for(m->dir_size), 4096)), dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size, (uint32_t*)nullptr, 0u,
                       (uint4*)nullptr, 0u, (uint2*)nullptr, (uint32_t*)nullptr, 0u, 0u);
    HIP_OK(hipGetLastError());
    uint32_t two[2] = {0, 0};
    HIP_OK(hipMemcpyAsync(two, &m->d_ctl->n_big, 8, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    m->far_rows_seen = two[0]; m->far_units_seen = std::max(two[1], 1u);
  }
  const uint32_t cap_rows = (uint32_t)std::min<uint64_t>(m->dir_used, (uint64_t)m->far_rows_seen * 5 / 4 + 4096);
  const uint32_t cap_units = (uint32_t)std::min<uint64_t>(m->arena.mapped >> (FAR_UNIT_LG + 3), (uint64_t)m->far_units_seen * 5 / 4 + 32768);
  uint32_t lg = 16;
  while (lg < 23 && (1ull << lg) < 4ull * est_nd + 2ull * cap_rows) lg++;
  m->far_unit_row.need(cap_units); m->far_unit_info.need(cap_units);
  const bool moved = m->far_tab.cap < ((size_t)1 << lg) || m->far_occ.cap < (size_t)cap_units * FAR_UNIT_WORDS || m->far_zeros.cap < cap_units;
  m->far_tab.need((size_t)1 << lg); m->far_occ.need((size_t)cap_units * FAR_UNIT_WORDS); m->far_zeros.need(cap_units);
  m->far_occ0.need((size_t)cap_units * FAR_UNIT_WORDS); m->far_clm.need((size_t)cap_units * FAR_UNIT_WORDS);      // (they move with far_occ)
  if (moved || m->far_tab_lg != lg) {
    struct { uint32_t mask; uint4* tab; const unsigned long long* occ; const uint32_t* zeros; } __attribute__((packed)) w = {(1u << lg) - 1u, m->far_tab.p, m->far_occ.p, m->far_zeros.p};
    static_assert(sizeof(w) == 28 && offsetof(ArenaHead, far_tab) == offsetof(ArenaHead, far_mask) + 4 && offsetof(ArenaHead, far_occ) == offsetof(ArenaHead, far_tab) + 8 &&
                  offsetof(ArenaHead, far_zeros) == offsetof(ArenaHead, far_occ) + 8, "ArenaHead layout");
    HIP_OK(hipMemcpyAsync(m->arena.base + offsetof(ArenaHead, far_mask), &w, sizeof(w), hipMemcpyHostToDevice, s));
    struct { const unsigned long long* occ0; uint32_t* clm; } w2 = {m->far_occ0.p, m->far_clm.p};
    static_assert(sizeof(w2) == 16 && offsetof(ArenaHead, far_clm) == offsetof(ArenaHead, far_occ0) + 8, "ArenaHead layout");
    HIP_OK(hipMemcpyAsync(m->arena.base + offsetof(ArenaHead, far_occ0), &w2, sizeof(w2), hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));                       // (`w` is on the stack; rare: the buffers moved)
    m->far_tab_lg = lg;
  }
  const uint32_t tmask = (1u << lg) - 1u;
  HIP_OK(hipMemsetAsync(m->far_tab.p, 0, (size_t)16 << lg, s));
  arena_head_set(m, offsetof(ArenaHead, far_overflow), 0u, s);
  HIP_OK(hipMemsetAsync(&m->d_ctl->n_big, 0, 8, s));       // n_big, n_units
  // (the rows of >= 2^FAR_ROW_LG cells: by the list k_grow_commit keeps once a pass over the directory has built it)
  m->big_list.need(BIG_LIST_CAP + 1);
  const bool by_list = m->big_list_valid;
  if (!by_list) HIP_OK(hipMemsetAsync(m->big_list.p, 0, 4, s));
  hipLaunchKernelGGL(k_far_rows, dim3(std::min<uint32_t>(blocks_for(by_list ? std::max<uint32_t>(m->far_rows_seen * 2u, 4096u) : m->dir_size), 4096)), dim3(256), 0, s, m->d_ctl,
                     m->d_dir, m->dir_size, m->far_unit_row.p, cap_units, m->far_tab.p, tmask, m->far_unit_info.p, m->big_list.p, BIG_LIST_CAP, by_list ? 1u : 0u);
  m->big_list_valid = true;
  DBG_STEP(m, s, "k_far_rows");
  m->far_bloom.need((size_t)1 << (FAR_BLOOM_LG - 5));
  HIP_OK(hipMemsetAsync(m->far_bloom.p, 0, (size_t)4 << (FAR_BLOOM_LG - 5), s));
  hipLaunchKernelGGL(k_far_keys, dim3(std::min<uint32_t>(blocks_for(est_nd), 4096)), dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, dl, x, y,
                     m->in_stride, m->far_tab.p, tmask, (1u << lg) / 4u, all_far ? 1u : 0u, m->far_bloom.p);
  DBG_STEP(m, s, "k_far_keys");
  const bool place = m->far_place;
  if (place) { m->far_rcnt.need((size_t)cap_units + 1); m->far_bucket.need((size_t)cap_units * FAR_BUCKET_PER_UNIT); m->far_prows.need(cap_rows); }
  hipLaunchKernelGGL(k_far_scan, dim3(std::min<uint32_t>(blocks_for((uint64_t)cap_units * 64), 32768)), dim3(256), 0, s, m->d_ctl, m->d_dir, m->far_unit_row.p, cap_units,
                     m->arena.base, m->far_tab.p, tmask, m->far_occ.p, m->far_zeros.p, m->far_occ0.p, m->far_clm.p, place ? m->far_rcnt.p : nullptr, m->far_bloom.p, m->far_unit_info.p);
  DBG_STEP(m, s, "k_far_scan");
  if (place) {
    // the keys the scan has not found are placed a row at a time (k_far_place) before the pass looks for them
    hipLaunchKernelGGL(k_far_absent, dim3(std::min<uint32_t>(blocks_for((uint64_t)1 << lg), 2048)), dim3(256), 0, s, m->d_dir, m->far_unit_row.p, m->arena.base, m->far_tab.p, tmask,
                       m->far_rcnt.p, cap_units, m->far_bucket.p, m->far_prows.p, cap_rows);
    hipLaunchKernelGGL((k_far_place<64, FAR_ROW_LG, FAR_PLACE_SMALL_LG>), dim3(2048), dim3(64), far_place_lds_bytes(FAR_PLACE_SMALL_LG), s, m->d_dir, m->far_unit_row.p, m->arena.base,
                       m->far_tab.p, m->far_rcnt.p, cap_units, m->far_bucket.p, m->far_prows.p, cap_rows, m->far_occ.p, m->far_zeros.p, m->far_occ0.p);
    hipLaunchKernelGGL((k_far_place<512, FAR_PLACE_SMALL_LG + 1, REST_LDS_MAX_LG>), dim3(256), dim3(512), far_place_lds_bytes(REST_LDS_MAX_LG), s, m->d_dir, m->far_unit_row.p, m->arena.base,
                       m->far_tab.p, m->far_rcnt.p, cap_units, m->far_bucket.p, m->far_prows.p, cap_rows, m->far_occ.p, m->far_zeros.p, m->far_occ0.p);
    DBG_STEP(m, s, "k_far_place");
  }
  HIP_OK(hipGetLastError());
  arena_head_set(m, offsetof(ArenaHead, far_on), 1u, s);
  return true;
}

template <int OP>
void launch_apply(Matrix* m, hipStream_t s, uint32_t n, const uint32_t* idx, const uint32_t* x,
                  const uint32_t* y, const uint32_t* v, uint32_t* out, uint32_t* defer) {
  if ((OP == OP_INCR || OP == OP_DECR) && idx && m->clustered && m->retry_far && n >= (1u << 15) && n <= (1u << 21) && m->in_stride != 3) {
    // a long retry list of a clustered table in a round the host drives (the young table's batches: rows that double twice leave
    // 10^5 ops for a third round, most of them new keys that wrap onto the runs of the hot rows): through the batch's far join,
    // rebuilt for this list -- claimed inserts by rank instead of a queue at one front per run.  (Measured and rejected as #51
    // while a claim could take a hundred turns; with claims by rank: rounds of 6.7 and 4.2 ms in the dense stream's second and
    // third batch.)
    HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&m->d_ctl->n_prev), (int)n, 1, s));      // (the join's kernels read the list's length there)
    if (far_join_enqueue(m, s, idx, x, y, n)) {
      hipLaunchKernelGGL((k_apply_wpo_far<OP == OP_DECR ? OP_DECR : OP_INCR>), dim3(65536), dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n,
                         idx, x, y, v, out, defer, m->in_stride);
      HIP_OK(hipGetLastError());
      arena_head_set(m, offsetof(ArenaHead, far_on), 0u, s);
      return;
    }
  }
  if (OP != OP_GET && idx && m->clustered && n <= m->wpo_max) {
    // a retry list of a clustered table: a wave per op (k_apply_wpo)
    hipLaunchKernelGGL((k_apply_wpo<OP>), dim3(std::min<uint32_t>(blocks_for((uint64_t)n * (64 / SMX_WPO_OPS)), 65536)), dim3(256), 0, s, m->d_ctl, m->d_dir,
                       m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer, m->in_stride);
    HIP_OK(hipGetLastError());
    return;
  }
  if (OP == OP_GET && m->d_hints && m->clustered && !idx && m->get_split)      // (clustered tables: a miss at home asks for the next cells and the hint together -- k_get_clu)
    hipLaunchKernelGGL(k_get_clu, dim3(blocks_for(n)), dim3(256), 0, s, m->d_dir, m->dir_size - 1, m->arena.base, n, x, y, out, m->in_stride);
  else if (m->d_hints)      // (the instantiation that asks the hint table after HINT_BUDGET cells: ArenaHead)
    hipLaunchKernelGGL((k_apply<OP, true>), dim3(blocks_for(n)), dim3(256), 0, s, m->d_ctl, m->d_dir,
                       m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer, m->in_stride);
  else
  hipLaunchKernelGGL((k_apply<OP>), dim3(blocks_for(n)), dim3(256), 0, s, m->d_ctl, m->d_dir,
                     m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer, m->in_stride);
  HIP_OK(hipGetLastError());
}

template <int OP>
void launch_apply_agg(Matrix* m, hipStream_t s, uint32_t n, const uint32_t* idx, const uint32_t* x,
                      const uint32_t* y, const uint32_t* v, uint32_t* out, uint32_t* defer) {
  // no result array from the caller: the instantiation that skips the results (and may take the paths that are exact in
  // the table but not in what they would have returned)
  const dim3 grid(blocks_for(n, AGG_TILE)), block(AGG_THREADS);
  if (m->no_ret) {
    if (m->in_stride == 3)
      hipLaunchKernelGGL((k_apply_agg<OP, 3, false>), grid, block, 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer);
    else
      hipLaunchKernelGGL((k_apply_agg<OP, 1, false>), grid, block, 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer);
  } else if (m->in_stride == 3) {
    hipLaunchKernelGGL((k_apply_agg<OP, 3>), grid, block, 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer);
  } else if (m->d_hints && m->clustered) {
    hipLaunchKernelGGL((k_apply_agg_clu<OP>), grid, block, 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer);
  } else {
    hipLaunchKernelGGL((k_apply_agg<OP>), grid, block, 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n, idx, x, y, v, out, defer);
  }
  HIP_OK(hipGetLastError());
}

void launch_apply_op(Matrix* m, int op, hipStream_t s, uint32_t n, const uint32_t* idx,
                     const uint32_t* x, const uint32_t* y, const uint32_t* v, uint32_t* out,
                     uint32_t* defer) {
  bool timed = m->profile && idx == nullptr;
  if (timed) HIP_OK(hipEventRecord(m->ev0, s));
  switch (op) {
    case OP_GET:  launch_apply<OP_GET>(m, s, n, idx, x, y, v, out, defer); break;
    case OP_SET:
      if (idx == nullptr && !m->long_probes && n >= m->agg_min) {
        // round 0 of a set batch: one winner per distinct key and tile (k_set_fold); the passes after the rounds run over
        // these entries
        const uint32_t tiles = blocks_for(n, AGG_TILE);
        m->set_entries = tiles * AGG_TILE;
        m->ent_idx.need(m->set_entries);
        if (m->in_stride == 3)
          hipLaunchKernelGGL((k_set_fold<3>), dim3(tiles), dim3(AGG_THREADS), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n, x, y, v, out, defer, m->ent_idx.p, m->cellp.p);
        else
          hipLaunchKernelGGL((k_set_fold<1>), dim3(tiles), dim3(AGG_THREADS), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, n, x, y, v, out, defer, m->ent_idx.p, m->cellp.p);
        HIP_OK(hipGetLastError());
      } else {
        launch_apply<OP_SET>(m, s, n, idx, x, y, v, out, defer);
      }
      break;
    case OP_INCR:
      if (!m->long_probes && n >= (idx && !m->retry_may_repeat ? m->agg_min_retry : m->agg_min)) launch_apply_agg<OP_INCR>(m, s, n, idx, x, y, v, out, defer);
      else launch_apply<OP_INCR>(m, s, n, idx, x, y, v, out, defer);
      break;
    case OP_DECR:
      if (!m->long_probes && n >= (idx && !m->retry_may_repeat ? m->agg_min_retry : m->agg_min)) launch_apply_agg<OP_DECR>(m, s, n, idx, x, y, v, out, defer);
      else launch_apply<OP_DECR>(m, s, n, idx, x, y, v, out, defer);
      break;
    default: smx_die("bad op code");
  }
  if (timed) HIP_OK(hipEventRecord(m->ev1, s));
}

void account_kernel_time(Matrix* m, int op, uint32_t n) {
  float ms = 0;
  HIP_OK(hipEventSynchronize(m->ev1));
  HIP_OK(hipEventElapsedTime(&ms, m->ev0, m->ev1));
  m->st.kernel_ms[op] += ms;
  m->st.kernel_launches[op]++;
  m->st.kernel_ops[op] += n;
}

// The get kernel is ONE asynchronous launch: waiting for its stop event right behind it (as the write path may, which
// has just read its counters back anyway) would turn every profiled get into a synchronous call and expose the next
// call's launch latency -- 0.1-0.3 ms per step on a slow host (round 3: the kernel spans of a bench step summed to
// 2.27 ms, its wall time was 2.6).  So get launches are timed with event PAIRS from a small ring that are read later:
// when the ring wraps (that launch finished long ago), at smatrix_stats() and when profiling is switched off.
void get_timing_resolve(Matrix* m, size_t keep) {
  while (m->get_pending.size() > keep) {
    Matrix::TimedLaunch t = m->get_pending.front();
    m->get_pending.pop_front();
    float ms = 0;
    HIP_OK(hipEventSynchronize(t.e1));
    HIP_OK(hipEventElapsedTime(&ms, t.e0, t.e1));
    m->st.kernel_ms[OP_GET] += ms;
    m->st.kernel_launches[OP_GET]++;
    m->st.kernel_ops[OP_GET] += t.n;
    m->ev_free.push_back(t.e0);
    m->ev_free.push_back(t.e1);
  }
}
hipEvent_t get_timing_event(Matrix* m) {
  if (m->ev_free.empty()) {
    hipEvent_t e;
    HIP_OK(hipEventCreate(&e));
    return e;
  }
  hipEvent_t e = m->ev_free.back();
  m->ev_free.pop_back();
  return e;
}

// nt / gu / nk: what prep has counted (read back), or -- `spec` -- the host's ESTIMATES for a round whose counters it has
// not read: every buffer is sized for them, the launches get grids for them (the kernels loop over the device-side
// counts), and k_grow_plan refuses what does not fit.
void grow_rows(Matrix* m, hipStream_t s, uint32_t nt, uint64_t gu, const uint32_t* nk, bool spec) {
  ensure_arena_free(m, gu, s);
  // chunk bounds: a table of 2^lg cells has max(1, 2^lg/64) chunks; units = 2^lg/16
  m->map_new.need((size_t)nt + gu / 4 + 1);
  m->map_old.need((size_t)nt + gu / 8 + 1);
  if (m->clustered && m->home_on && m->rest_lds) {
    // k_grow_rest_lds' time slices: a mask of displaced cells per old chunk, and the table of slices -- at most one per
    // rest_slice_cells displaced cells of a row plus one per row, whatever the plan accepts of the chunks there is room for
    m->disp_mask.need(m->map_old.cap);
    m->rest_tab.need(2 + 2 * ((size_t)m->map_old.cap * 64 / m->rest_slice_cells + m->tasks.cap + 1));
  }
  uint64_t cap_units = m->arena.mapped / UNIT_BYTES;
  if (spec && m->spec_tiny) cap_units = std::min<uint64_t>(cap_units, m->arena_next + gu);   // (tests: arena refusals too)
  for (uint32_t c = 0; c < N_CLASSES; c++) ensure_free_cap(m, c, nt, s);     // every task retires one block
  // (spec: the arena holds gu units beyond the host's mirror of the bump pointer -- row creation in prep may have taken
  //  some of the slack ensure_arena_free was asked for, hence the cap is what is mapped, checked per allocation)
  // the keys that wait for these doublings (k_prep's records): buckets by the plan, grouped right behind it
  const bool pend = m->pend_armed && m->clustered && m->home_on;
  m->pend_armed = false;
  uint32_t pend_hash_lg = 16;
  if (pend) {
    m->pend_keys.need(std::min<uint64_t>(8ull * nt + 4ull * gu + 1024, 0x7FFFFFFFull));
    m->task_of.need(m->dir_size);
    while (pend_hash_lg < 24 && (1ull << pend_hash_lg) < 2ull * m->pend_est) pend_hash_lg++;
    m->pend_hash.need((size_t)1 << pend_hash_lg);
  }
  hipLaunchKernelGGL(k_grow_plan, dim3(std::min<uint32_t>(blocks_for(nt), 1024)), dim3(256), 0, s,
                     m->d_ctl, m->tasks.p, cap_units, m->fl, spec ? nt : 0xFFFFFFFFu,
                     spec ? (uint32_t)std::min<uint64_t>(m->map_old.cap, m->map_new.cap / 2) : 0xFFFFFFFFu,
                     pend ? m->pend_ctl.p : nullptr, (uint32_t)m->pend_keys.cap, m->task_of.p);
  DBG_STEP(m, s, "k_grow_plan");
  if (pend) {
    HIP_OK(hipMemsetAsync(m->pend_hash.p, 0, (size_t)8 << pend_hash_lg, s));
    hipLaunchKernelGGL(k_pend_group, dim3(std::min<uint32_t>(blocks_for(std::max<uint32_t>(m->pend_est, 1)), 2048)), dim3(256), 0, s, m->d_ctl, m->tasks.p, m->task_of.p,
                       m->pend_rec.p, (uint32_t)m->pend_rec.cap, m->pend_ctl.p, m->pend_hash.p, (1u << pend_hash_lg) - 1u, m->pend_keys.p);
    DBG_STEP(m, s, "k_pend_group");
  }
  const uint32_t* pend_keys = pend ? m->pend_keys.p : nullptr;
  const uint32_t n_chunked = nk[GROW_CHUNKED];
  // the chunked passes of the large rows touch other rows than the in-LDS rehashes: they run on a helper stream beside
  // them, from the plan on (non-blocking stream + events: the caller's stream may be the legacy default stream, which a
  // blocking helper would serialise with -- round 1's attempt with plain streams was erratic).  Same box, 3 runs each:
  // 2.72 -> 2.65 ms per config-2 step.
  const bool fork = m->grow_fork && n_chunked && (nk[0] || nk[1] || nk[2]);
  if (fork) HIP_OK(hipEventRecord(m->ev_fork, s));
  // (a clustered matrix's rebuilds take the waiting keys in: their own instantiation -- the scrambled stream's launches stay round 5's code)
  auto lds_kinds = [&](auto pend_c) {
    constexpr bool P = decltype(pend_c)::value;
    if (nk[0])
      hipLaunchKernelGGL((k_grow_lds<64, GROW_LG0, P>), dim3(std::min<uint32_t>(nk[0], 32768)), dim3(64), grow_lds_bytes(GROW_LG0), s,
                         m->d_ctl, m->tasks.p, m->klist.p, 0u, m->arena.base, pend_keys);
    if (nk[1])
      hipLaunchKernelGGL((k_grow_lds<256, GROW_LG1, P>), dim3(std::min<uint32_t>(nk[1], 4096)), dim3(256), grow_lds_bytes(GROW_LG1), s,
                         m->d_ctl, m->tasks.p, m->klist.p + m->klist_cap, 1u, m->arena.base, pend_keys);
    if (nk[2])
      hipLaunchKernelGGL((k_grow_lds<1024, GROW_LG2, P>), dim3(std::min<uint32_t>(nk[2], 1024)), dim3(1024), grow_lds_bytes(GROW_LG2), s,
                         m->d_ctl, m->tasks.p, m->klist.p + 2 * (size_t)m->klist_cap, 2u, m->arena.base, pend_keys);
  };
  if (pend_keys) lds_kinds(std::true_type{}); else lds_kinds(std::false_type{});
  DBG_STEP(m, s, "k_grow_lds x3");
  const uint64_t oc_bound = (uint64_t)n_chunked + gu / 8, nc_bound = (uint64_t)n_chunked + gu / 4;
  hipStream_t sc = s;
  if (fork) {
    HIP_OK(hipStreamWaitEvent(m->helper, m->ev_fork, 0));
    sc = m->helper;
  }
  if (n_chunked) {
    // clustered rows (a batch has shown long probe sequences: dense ids): the move in two passes, at-home cells first
    // (smx_kernels.hpp "clustered rows")
    // (scrambled ids, same box, A/B: the two passes cost 2.475 / 2.487 ms per step against 2.447 / 2.442 for the single one)
    // (the first pass leaves the new tables' at-home bitmaps behind their blocks: the second pass, the duplicate check and,
    //  from then on, the op kernels' long probes step over at-home cells by them -- smx_kernels.hpp HOME_LG)
    const bool two_pass = m->clustered && m->home_on;
    if (two_pass)
      hipLaunchKernelGGL(k_grow_map<true>, dim3(std::min<uint32_t>(std::max<uint32_t>(n_chunked, 1), 2048)),
                         dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->klist.p + 3 * (size_t)m->klist_cap, m->map_old.p, m->map_new.p, m->arena.base);
    else
      hipLaunchKernelGGL(k_grow_map<false>, dim3(std::min<uint32_t>(std::max<uint32_t>(n_chunked, 1), 2048)),
                         dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->klist.p + 3 * (size_t)m->klist_cap, m->map_old.p, m->map_new.p, nullptr);
    if (two_pass) {
      hipLaunchKernelGGL(k_grow_move_home, dim3(std::min<uint32_t>(blocks_for(oc_bound * 64), 16384)),
                         dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->map_old.p, m->arena.base, m->rest_lds ? m->disp_mask.p : nullptr);
      // the displaced cells: a workgroup per row on an occupancy bitmap in LDS (k_grow_rest_lds); the chunked pass behind it
      // takes only what that kernel leaves (giant rows, rows with wrapped cells).  SMATRIX_REST_LDS=0: the chunked pass alone
      if (m->rest_lds) {
        hipLaunchKernelGGL(k_grow_rest_count, dim3(std::min<uint32_t>(blocks_for(oc_bound), 1024)), dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->map_old.p, m->disp_mask.p);
        hipLaunchKernelGGL(k_grow_rest_plan, dim3(1), dim3(1024), 0, sc, m->d_ctl, m->tasks.p, m->klist.p + 3 * (size_t)m->klist_cap, m->rest_tab.p,
                           (uint32_t)std::min<size_t>((m->rest_tab.cap - 2) / 2, 0x7FFFFFFFu), m->rest_slice_cells);
        hipLaunchKernelGGL(k_grow_rest_lds, dim3(m->rest_grid), dim3(REST_THREADS), rest_lds_bytes(), sc,
                           m->d_ctl, m->tasks.p, m->rest_tab.p, m->disp_mask.p, m->arena.base, m->st.batches >= m->rest_dbg_from ? m->rest_dbg : nullptr, m->rest_dbg_mode | ((uint32_t)(m->st.batches & 31u) << 8), pend_keys);
      }
      hipLaunchKernelGGL(k_grow_move_rest, dim3(std::min<uint32_t>(blocks_for(oc_bound * 64), 16384)),
                         dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->map_old.p, m->arena.base, m->rest_lds);
    } else {
      hipLaunchKernelGGL(k_grow_move, dim3(std::min<uint32_t>(blocks_for(oc_bound * 64), 16384)),
                         dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->map_old.p, m->arena.base);
    }
    hipLaunchKernelGGL(k_grow_finish, dim3(std::min<uint32_t>(blocks_for(nc_bound * 64), 16384)),
                       dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->map_new.p, m->arena.base, two_pass);
    hipLaunchKernelGGL(k_grow_zero, dim3(std::min<uint32_t>(blocks_for(oc_bound * 64), 16384)),
                       dim3(256), 0, sc, m->d_ctl, m->tasks.p, m->map_old.p, m->arena.base);
  }
  DBG_STEP(m, s, "chunked passes");
  if (fork) {
    HIP_OK(hipEventRecord(m->ev_join, sc));
    HIP_OK(hipStreamWaitEvent(s, m->ev_join, 0));
  }
  hipLaunchKernelGGL(k_grow_commit, dim3(std::min<uint32_t>(blocks_for(nt), 1024)), dim3(256), 0, s,
                     m->d_ctl, m->tasks.p, m->d_dir, m->arena.base, m->fl, m->big_list_valid ? m->big_list.p : nullptr, BIG_LIST_CAP);
  HIP_OK(hipGetLastError());
  DBG_STEP(m, s, "k_grow_commit");
  if (m->trace_rounds && nt > 1000 && !spec) {      // who grows?  (cells moved, by log2 of the old row size)
    std::vector<GrowTask> ht(nt);
    HIP_OK(hipMemcpyAsync(ht.data(), m->tasks.p, (size_t)nt * sizeof(GrowTask), hipMemcpyDeviceToHost, s));
    sync_counted(m, s);
    uint64_t cnt[32] = {0};
    for (const GrowTask& k : ht) cnt[k.old_lg & 31]++;
    fprintf(stderr, "[smatrix]   growth by old size:");
    for (int lg = 4; lg < 32; lg++)
      if (cnt[lg]) fprintf(stderr, " 2^%d x%llu (%.1fM cells)", lg, (unsigned long long)cnt[lg], cnt[lg] * (double)(1ull << lg) / 1e6);
    fprintf(stderr, "\n");
  }
  m->arena_next += gu;   // upper bound until the next readback (recycled blocks take nothing from the arena)
  if (!spec) m->st.rows_grown += nt;
}

// The bulk path of a write batch (smx_kernels.hpp "the bulk path"): `nd` deferred ops in `dl`; returns how many were
// handed back (in `dl_out`) for the round loop.
template <int OP>
uint32_t run_bulk_t(Matrix* m, uint32_t nd, const uint32_t* dl, uint32_t* dl_out, const uint32_t* x, const uint32_t* y,
                    const uint32_t* v, uint32_t* out, hipStream_t s) {
  // 1. the rows (directory growth included) -- prep without its growth decisions
  // (round 4) a list that may not fit the directory: its missing rows are counted first and the directory is sized ONCE
  // (k_fix_count_rows), instead of running the creation pass into "directory full" once per factor of four
  // (only when the list dwarfs the directory -- the first batches of a matrix; a bulk load in progress, whose directory has
  //  grown with its rows, keeps the cheap path: one creation pass, now and then a rebuild)
  bool created_from_set = false;
  // (the scratch set is capped at 2^28 slots, so lists that may name more than 2^27 distinct rows keep the
  //  one-factor-at-a-time growth -- a set that fills up would never let k_fix_count_rows' probe end)
  if (nd >= m->fix_presize_min && nd / 16 >= m->dir_size && nd <= (1u << 27)) {
    uint64_t slots = 1u << 16;
    while (slots < 2ull * std::min<uint64_t>(nd, 1ull << 27)) slots <<= 1;
    m->cold_set.need_on(slots, s);                    // (kept: the cold start's key set is the same size; run_write releases it)
    zero_async(m->cold_set.p, slots * 8, s);
    HIP_OK(hipMemsetAsync(m->d_small + 12, 0, 8, s));
    hipLaunchKernelGGL(k_fix_count_rows, dim3(std::min<uint32_t>(blocks_for(nd, 256 * FIXR_OPT), 4096)), dim3(256), 0, s, m->d_dir, m->dir_size - 1, nd, dl,
                       x, m->in_stride, m->cold_set.p, slots - 1, m->d_small + 12);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(m->h_small + 12, m->d_small + 12, 8, hipMemcpyDeviceToHost, s));
    sync_counted(m, s);
    const uint64_t need = (uint64_t)m->dir_used + m->h_small[12] + m->h_small[13];
    // the size the old one-factor-at-a-time growth would have ended at: x4 while the rows do not fit half the slots, x2
    // when they fill half of them afterwards
    uint64_t size = m->dir_size;
    while (need > size / 2) size *= 4;
    if (need * 2 >= size) size *= 2;
    if (m->trace_rounds)
      fprintf(stderr, "[smatrix] batch %llu bulk path: %u ops name %u rows the directory lacks; directory %u -> %llu slots\n",
              (unsigned long long)m->st.batches, nd, m->h_small[12] + m->h_small[13], m->dir_size, (unsigned long long)size);
    if (size > m->dir_size) grow_directory(m, (uint32_t)(size / m->dir_size), s);
    // ... and the rows are created from the set itself (k_fix_create_set): a sweep over its slots instead of a second fold of
    // the whole list.  The creation pass below then finds every row in place -- it is kept for the one id the set cannot
    // hold (0xFFFFFFFF) and runs only when that id was seen.
    if (need > m->dir_used && need <= m->dir_size / 2) {
      ensure_arena_free(m, need - m->dir_used, s);
      hipLaunchKernelGGL(k_fix_create_set, dim3((uint32_t)std::min<uint64_t>(blocks_for(slots, 256 * FIXS_OPT), 1024)), dim3(256), 0, s, m->d_ctl, m->d_dir,
                         m->dir_size - 1, m->cold_set.p, slots, (uint64_t)(m->arena.mapped / UNIT_BYTES), m->fl);
      HIP_OK(hipGetLastError());
      created_from_set = true;                               // (k_fix_create leaves the id 0xFFFFFFFF to the round loop's prep as well)
      // the grouping passes' buffers, allocated while the sweep runs
      m->fx_where.need(nd); m->fx_grouped.need(nd); m->fx_rank.need(nd);
      const uint32_t rows_guess = (uint32_t)std::min<uint64_t>(nd, need);
      m->fx_touched.need(std::max<uint32_t>(rows_guess, 1)); m->fx_excl.need(std::max<uint32_t>(rows_guess, 1));
      m->fx_tiles.need((rows_guess + SCAN_TILE - 1) / SCAN_TILE + 2);
      if (m->fx_dir_size != m->dir_size) { m->fx_cnt.need(m->dir_size); m->fx_cur.need(m->dir_size); m->fx_pos.need(m->dir_size); }
      ctl_read(m, s);
      if (m->h_ctl->arena_oom) smx_die("internal: arena reservation too small");
    }
  }
  for (int tries = 0; !created_from_set; tries++) {
    if (tries > 40) smx_die("bulk path: the directory does not take the batch's rows");
    const uint32_t dir_limit = m->dir_size / 2;
    const uint32_t room = dir_limit > m->dir_used ? dir_limit - m->dir_used : 0;
    ensure_arena_free(m, std::min<uint64_t>(nd, room), s);
    ctl_reset_round(m, s);
    hipLaunchKernelGGL(k_fix_create, dim3(std::min<uint32_t>(blocks_for(nd, 256 * FIXR_OPT), 4096)), dim3(256), 0, s, m->d_ctl, m->d_dir,
                       m->dir_size - 1, dir_limit, nd, dl, x, m->in_stride, (uint64_t)(m->arena.mapped / UNIT_BYTES), m->fl);
    HIP_OK(hipGetLastError());
    if (tries == 0) {
      // (round 4) the grouping passes' buffers are allocated WHILE the creation pass runs -- a dozen hipMallocs, 0.15 ms of an
      // idle GPU when they came behind the read-back (they are sized for the op count, or grow again with the directory)
      m->fx_where.need(nd); m->fx_grouped.need(nd); m->fx_rank.need(nd);
      const uint32_t rows_guess = (uint32_t)std::min<uint64_t>(nd, (uint64_t)m->dir_size / 2);
      m->fx_touched.need(std::max<uint32_t>(rows_guess, 1)); m->fx_excl.need(std::max<uint32_t>(rows_guess, 1));
      m->fx_tiles.need((rows_guess + SCAN_TILE - 1) / SCAN_TILE + 2);
      if (m->fx_dir_size != m->dir_size) { m->fx_cnt.need(m->dir_size); m->fx_cur.need(m->dir_size); m->fx_pos.need(m->dir_size); }
    }
    ctl_read(m, s);
    if (m->h_ctl->arena_oom) smx_die("internal: arena reservation too small");
    const bool full = m->h_ctl->dir_full != 0;
    if (full) grow_directory(m, 4, s);
    else if ((uint64_t)m->dir_used * 2 >= m->dir_size) grow_directory(m, 2, s);
    if (!full) break;
  }
  // 2. ops per row, scan, scatter.  cnt / cursor are indexed by directory slot and ALL ZERO between batches (k_fix_rows
  //    clears what it takes), so nothing here costs O(directory): the passes run over the batch's ops and its touched rows
  const uint32_t ds = m->dir_size;
  if (m->fx_dir_size != ds) {
    m->fx_cnt.need(ds); m->fx_cur.need(ds); m->fx_pos.need(ds);
    HIP_OK(hipMemsetAsync(m->fx_cnt.p, 0, (size_t)ds * 4, s));
    HIP_OK(hipMemsetAsync(m->fx_cur.p, 0, (size_t)ds * 4, s));
    m->fx_dir_size = ds;
  }
  const uint32_t rows_max = (uint32_t)std::min<uint64_t>(nd, m->dir_used);          // touched rows at most
  const uint32_t ntiles = (rows_max + SCAN_TILE - 1) / SCAN_TILE;
  // (the stacks of retired blocks get their room NOW, while the stream is idle: a stack that has to grow copies its
  //  contents and waits for the stream -- between the scatter pass and the row pass that was 0.1 ms of idle GPU)
  for (uint32_t c = 0; c <= FIX_MAX_LG - ROW_FIRST_LG; c++) ensure_free_cap(m, c, rows_max, s);
  m->fx_touched.need(std::max<uint32_t>(rows_max, 1)); m->fx_excl.need(std::max<uint32_t>(rows_max, 1)); m->fx_tiles.need(ntiles + 2);
  m->fx_where.need(nd); m->fx_grouped.need(nd); m->fx_rank.need(nd);
  ctl_reset_round(m, s);                                       // n_defer: what is handed back; n_tasks: rows touched
  hipLaunchKernelGGL(k_fix_count, dim3(std::min<uint32_t>(blocks_for(nd, 256 * FIXC_OPT), 8192)), dim3(256), 0, s, m->d_ctl, m->d_dir, ds - 1, nd,
                     dl, x, y, m->in_stride, m->fx_cnt.p, m->fx_where.p, dl_out, m->fx_touched.p, m->fx_pos.p, m->fx_rank.p);
  HIP_OK(hipMemsetAsync(m->fx_tiles.p + ntiles, 0, 16, s));     // {total, "rows of the wide class exist"}
  hipLaunchKernelGGL(k_fix_scan_tiles, dim3(std::max(ntiles, 1u)), dim3(256), 0, s, m->d_ctl, m->d_dir, m->fx_cnt.p, m->fx_touched.p,
                     m->fx_excl.p, m->fx_tiles.p, m->fx_tiles.p + ntiles + 1);
  hipLaunchKernelGGL(k_fix_scan_tops, dim3(1), dim3(1024), 0, s, m->fx_tiles.p, ntiles, m->fx_tiles.p + ntiles);
  hipLaunchKernelGGL(k_fix_scan_add, dim3(blocks_for(std::max<uint32_t>(rows_max, 1))), dim3(256), 0, s, m->d_ctl, m->fx_excl.p, m->fx_tiles.p);
  hipLaunchKernelGGL(k_fix_scatter, dim3(std::min<uint32_t>(blocks_for(nd, 256 * FIXC_OPT), 8192)), dim3(256), 0, s, m->d_ctl, m->d_dir, m->fx_cnt.p, nd, dl, m->fx_where.p, m->fx_excl.p,
                     m->fx_pos.p, m->fx_rank.p, m->fx_grouped.p, dl_out);
  HIP_OK(hipGetLastError());
  uint64_t tw[2] = {0, 0};
  HIP_OK(hipMemcpyAsync(tw, m->fx_tiles.p + ntiles, 16, hipMemcpyDeviceToHost, s));
  sync_counted(m, s);
  const uint64_t units = tw[0] >> 32;
  // 3. the new blocks: ONE reservation for all rows (their offsets are the scan's upper halves)
  ensure_arena_free(m, units, s);
  if (m->arena_next + units >= (1ull << 32)) smx_die("row arena exhausted");
  const uint64_t new_base0 = m->arena_next;
  const dim3 fgrid(std::max<uint32_t>(1, std::min<uint32_t>((rows_max + FIX_WAVES - 1) / FIX_WAVES, 8192)));   // (no row at all: the one id this path leaves to prep)
  hipLaunchKernelGGL((k_fix_rows<OP, FIX_MAX_LG - 1>), fgrid, dim3(64 * FIX_WAVES), 0, s,
                     m->d_ctl, m->d_dir, m->fx_touched.p, m->arena.base, m->fx_cnt.p, m->fx_cur.p, m->fx_excl.p, m->fx_grouped.p, y, v,
                     m->in_stride, out, dl_out, new_base0, m->fl);
  if (tw[1])
    hipLaunchKernelGGL((k_fix_rows<OP, FIX_MAX_LG>), fgrid, dim3(64 * FIX_WAVES), 0, s,
                     m->d_ctl, m->d_dir, m->fx_touched.p, m->arena.base, m->fx_cnt.p, m->fx_cur.p, m->fx_excl.p, m->fx_grouped.p, y, v,
                     m->in_stride, out, dl_out, new_base0, m->fl);
  HIP_OK(hipGetLastError());
  const uint64_t next = new_base0 + units;
  HIP_OK(hipMemcpyAsync(&m->d_ctl->arena_next, &next, 8, hipMemcpyHostToDevice, s));
  std::vector<void*> put_aside;
  {
    // (round 4) what the round loop and a cold start will ask for next, allocated while the row passes run: task lists for
    // the handed-back ops, the packed-key buffer of the distinct pending keys
    // (a buffer that has to grow must not hipFree its old block here -- hipFree waits for the row passes: the old blocks are
    //  put aside and freed behind the read-back)
    const uint64_t bound = std::min<uint64_t>(nd, m->dir_size);
    const auto grow = [&](auto& buf, size_t n) {
      if (n <= buf.cap) return;
      if (buf.p) put_aside.push_back(buf.p);
      buf.p = nullptr; buf.cap = 0;
      buf.need(n);
    };
    grow(m->tasks, bound); grow(m->klist, 4 * bound); grow(m->rebal, bound);
    if (m->cold_min && nd >= m->cold_min) {
      uint64_t slots = 1;
      while (slots < 2ull * nd) slots <<= 1;
      m->cold_set.need_on(slots, s);
      m->cold_keys[0].need_on(nd, s);
      m->cold_keys[1].need_on(nd / 4 + nd / 8, s);      // (what the first round leaves: usually a quarter of the list names distinct keys)
      if (m->small_first) m->cold_keys[2].need_on(nd / 4 + nd / 8, s);     // (dense ids: the walkers' list; the far join's index lists and zeroed amounts)
      if (m->small_first && m->cold_far && m->far_join) {
        m->cold_idx[0].need_on(m->cold_far_max, s); m->cold_idx[1].need_on(m->cold_far_max, s); m->cold_zero.need_on(3 * (size_t)m->cold_far_max, s);
      }
    }
  }
  ctl_read(m, s);
  m->put_aside.insert(m->put_aside.end(), put_aside.begin(), put_aside.end());      // (freed at the end of the batch)
  m->st.bulk_rounds++;
  m->st.bulk_ops += nd - m->h_ctl->n_defer;
  return m->h_ctl->n_defer;
}

// The write-batch round loop (device pointers).
//
// Steady state has a fixed shape: round 0 defers ~1 % of the batch (new keys of rows that stand at the reference's
// threshold), prep flags those rows, they double, round 1 applies the deferred ops, nothing is left.  Driven from the
// host that is two read-backs and a dozen small launches issued one by one behind the first read-back -- the GPU waits
// for each launch packet (round 3: the kernels between the end of round 0 and the get kernel add up to 0.30 ms, the
// span is 0.42-0.55 ms).  When the previous batch had that shape the whole sequence is therefore enqueued AT ONCE
// ("speculative chain"): op kernel, prep, every growth pass with grids and buffers sized from the previous batch's
// counts (x2), k_round_advance, the retry over the device-side list, prep again -- and ONE read-back.  Nothing is
// guessed about the data: the kernels loop over the device-side counts, and a growth task that does not fit the
// estimates is refused by k_grow_plan (its row stays as it is, its ops stay deferred), so whatever is left after the
// chain -- refused tasks, rows that double twice in one batch, a full directory -- is finished by the host-driven loop
// below exactly as before.  SMATRIX_SPEC=0 switches the chain off.
// Cold start of hot rows.  A brand-new row that ends a batch at 2^k cells needs one round per doubling (the reference
// doubles inline, src/smatrix.c:346-348 -- and every table between 16 and 2^k cells has to be filled to its threshold
// and re-inserted in slot order for the final layout to be one the reference can produce), and every such round used
// to re-run the op kernel over ALL ops still pending: the first batch of config 2 took 16 rounds of 1.9 ms over ~9 M
// ops of which a few hundred thousand could insert.  Now a large deferred list is reduced ONCE to one representative
// op per distinct key (k_dedup_keys), the rounds run over these with the lane-per-op kernel as inserts of {y, 0}
// (k_insert_keys: an incr by 0) -- same prep, same growth passes, same thresholds -- and the caller's list then runs
// once more over a table in which every key exists: all hits, with the real values and results.  That is the serial
// order "first the inserting incr by 0 of every new key, then the batch's ops".
// Returns false when the list is not worth it (few duplicates).
bool insert_pending_keys(Matrix* m, const uint32_t* list, uint32_t n_list, const uint32_t* x, const uint32_t* y, hipStream_t s) {
  // 1. distinct keys
  uint64_t slots = 1;
  while (slots < 2ull * n_list) slots <<= 1;
  m->cold_set.need_on(slots, s);
  m->cold_keys[0].need_on(n_list, s);
  zero_async(m->cold_set.p, slots * 8, s);
  HIP_OK(hipMemsetAsync(m->d_small + 12, 0, 8, s));      // (words 0..9 of the scratch belong to the scalar path and the partition)
  hipLaunchKernelGGL(k_dedup_keys, dim3(std::min<uint32_t>(blocks_for(n_list, DEDUP_THREADS * DEDUP_TRIPS), 4096)), dim3(DEDUP_THREADS), 0, s,
                     n_list, list, x, y, m->in_stride, m->cold_set.p, slots - 1, m->cold_keys[0].p, m->d_small + 12);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(m->h_small + 12, m->d_small + 12, 8, hipMemcpyDeviceToHost, s));
  sync_counted(m, s);
  uint32_t cur_n = m->h_small[12];
  // dense ids: a quarter of the distinct keys below the length of the list (scrambled ids: one in 2^32 / n_list)
  const bool dense_keys = (uint64_t)m->h_small[13] * 4 >= cur_n;
  if (m->trace_rounds)
    fprintf(stderr, "[smatrix] batch %llu cold start: %u pending ops name %u distinct keys, %u of them below %u%s\n", (unsigned long long)m->st.batches, n_list, cur_n,
            m->h_small[13], n_list, dense_keys && m->small_first ? ": the rounds take the keys below their row's size first" : "");
  // (few duplicates: not worth it, dense ids or not -- the second batch of the dense-id stream, 2.5 M distinct keys in 2.7 M pending
  //  ops, takes 22 ms through the rounds below against 14 through the op kernels'; the scratch stays while the table is young: run_write)
  if ((uint64_t)cur_n * 4 > (uint64_t)n_list * 3) return false;
  m->st.cold_starts++;
  m->st.cold_keys += cur_n;
  // dense ids ARE clustered tables as soon as rows are big: the mode is switched on here (hint table, at-home bitmaps, two-pass growth
  // of big rows, the walks of the rounds below by the far join) instead of after the first rounds full of long probes -- with the
  // keys below a row's size going in first there are few of those to see (the mode goes again when batches stay quiet: run_write)
  const bool was_clustered = m->clustered;      // (... when this batch's op rounds ran)
  if (dense_keys && m->small_first && !m->clustered_forced && !m->clustered) {
    m->clustered = true; m->clustered_quiet = 0;
    clustered_sync(m, s);
    if (m->trace_rounds) fprintf(stderr, "[smatrix]   dense ids: clustered tables from here on\n");
  }
  // a clustered table: the list may name keys that EXIST (what the pass in front of prep left for the retry: k_insert_keys,
  // INS_DROP_EXISTING) -- they leave it here, once, before the rounds that take every listed key for absent
  uint32_t cur_buf = 0;                                                // (which of the two buffers holds the round's input)
  if (was_clustered && m->dir_used) {
    m->cold_keys[1].need_on(cur_n, s);
    ctl_reset_round(m, s);
    const uint32_t wpo = cur_n <= (1u << 21) ? 1u : 0u;
    hipLaunchKernelGGL(k_insert_keys, dim3(wpo ? std::min<uint32_t>(blocks_for((uint64_t)cur_n * 64, INS_THREADS), 16384) : blocks_for(cur_n, INS_THREADS)), dim3(INS_THREADS), 0, s,
                       m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, cur_n, m->cold_keys[0].p, m->cold_keys[1].p, wpo, INS_DROP_EXISTING, (unsigned long long*)nullptr);
    HIP_OK(hipGetLastError());
    ctl_read(m, s);
    if (m->trace_rounds) fprintf(stderr, "[smatrix]   %u of the %u keys exist already (left by the pass in front of prep): dropped from the list\n", cur_n - m->h_ctl->n_defer, cur_n);
    cur_n = m->h_ctl->n_defer;
    cur_buf = 1;
    if (cur_n == 0) return true;
  }
  // 2. the rounds, over the keys (packed, x << 32 | y: every round streams its input and writes what stays deferred the same way)
  const unsigned long long* kin = m->cold_keys[cur_buf].p;
  uint32_t stalled = 0, rows_before = m->dir_used;
  for (uint32_t round = 0; cur_n; round++) {
    if (stalled > 8) smx_die("write batch did not converge (corrupt row table?)");
    const uint32_t dir_limit = m->dir_size / 2;
    const uint32_t room = dir_limit > m->dir_used ? dir_limit - m->dir_used : 0;
    m->tasks.need(std::min<uint64_t>(cur_n, m->dir_size));
    m->klist_cap = (uint32_t)std::min<uint64_t>(cur_n, m->dir_size);
    m->klist.need(4 * (size_t)m->klist_cap);
    m->rebal.need(std::min<uint64_t>(cur_n, m->dir_size));
    // (two buffers take turns: the dedup's output, sized for the whole list, and one for the first round's survivors)
    unsigned long long* other = cur_buf ? m->cold_keys[0].p : (m->cold_keys[1].need_on(cur_n, s), m->cold_keys[1].p);
    ensure_arena_free(m, std::min<uint64_t>(cur_n, room), s);
    ctl_reset_round(m, s);
    // (clustered tables, lists up to 2^21 keys: a wave per key -- k_insert_keys)
    const uint32_t ins_wpo = m->clustered && cur_n <= (1u << 21) ? 1u : 0u;
    unsigned long long* kout = other;
    if ((m->clustered || dense_keys) && m->small_first) {
      // clustered tables / dense ids: the keys below their row's size first, then any key whose home cell is free -- which also
      // sets the WALKERS aside (home cell taken, row not full) --, then the walks over those alone (k_insert_keys: mode).  The
      // round's list ends up in its input buffer again.
      unsigned long long* mine = const_cast<unsigned long long*>(kin);
      m->cold_keys[2].need_on(cur_n, s);
      unsigned long long* walk = m->cold_keys[2].p;
      hipLaunchKernelGGL(k_insert_keys, dim3(blocks_for(cur_n, INS_THREADS)), dim3(INS_THREADS), 0, s,
                         m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, cur_n, kin, other, 0u, INS_SMALL_ONLY, (unsigned long long*)nullptr);
      hipLaunchKernelGGL(k_list_advance, dim3(1), dim3(1), 0, s, m->d_ctl);
      hipLaunchKernelGGL(k_insert_keys, dim3(blocks_for(cur_n, INS_THREADS)), dim3(INS_THREADS), 0, s,
                         m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, cur_n, other, mine, 0u, INS_HOME_ONLY | INS_FROM_PREV, walk);
      hipLaunchKernelGGL(k_walk_advance, dim3(1), dim3(1), 0, s, m->d_ctl);
      kout = mine;
      bool far_joined = false;
      if (m->clustered && m->home_on && m->far_join && m->cold_far && cur_n <= m->cold_far_max) {
        // The walks by the batch's far join (the pass in front of prep of a steady batch, k_apply_wpo_far): the walkers are ops
        // "incr by 0" -- x and y the two words of a packed key, the amounts a zeroed array, the list 0..n-1.  New keys CLAIM their
        // cell by rank in the occupancy words instead of walking to the end of a run and queueing there (the wave-per-key launch
        // took 12 + 19 + 17 ms in the last three rounds of the dense stream's first batch: a few thousand keys per hot row at ONE
        // front, one winner per compare-and-swap).  What the pass leaves deferred it appends, as indices, behind the round's list.
        m->cold_idx[0].need_on(cur_n, s); m->cold_idx[1].need_on(cur_n, s); m->cold_zero.need_on(3 * (size_t)cur_n, s);
        zero_async(m->cold_zero.p, 8 * (size_t)cur_n, s);
        hipLaunchKernelGGL(k_iota, dim3(std::min<uint32_t>(blocks_for(cur_n), 4096)), dim3(256), 0, s, m->cold_idx[0].p, cur_n);
        const uint32_t* xk = reinterpret_cast<const uint32_t*>(walk) + 1, * yk = reinterpret_cast<const uint32_t*>(walk);
        const uint32_t stride_was = m->in_stride;
        m->in_stride = 2;
        far_joined = far_join_enqueue(m, s, m->cold_idx[0].p, xk, yk, cur_n, m->cold_all_far);
        if (far_joined) {
          hipLaunchKernelGGL((k_apply_wpo_far<OP_INCR>), dim3(65536), dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu,
                             m->cold_idx[0].p, xk, yk, m->cold_zero.p, m->cold_zero.p + 2 * (size_t)cur_n, m->cold_idx[1].p, 2u);     // (amounts: stride 2 like the keys, all zero; results: dense, behind them, nobody reads them)
          arena_head_set(m, offsetof(ArenaHead, far_on), 0u, s);     // (rows are about to double: the join's view of the tables ends here)
          hipLaunchKernelGGL(k_gather_keys, dim3(std::min<uint32_t>(blocks_for(cur_n), 4096)), dim3(256), 0, s, m->d_ctl, m->cold_idx[1].p, walk, kout, cur_n);
        }
        m->in_stride = stride_was;
      }
      if (!far_joined)
        hipLaunchKernelGGL(k_insert_keys, dim3(ins_wpo ? std::min<uint32_t>(blocks_for((uint64_t)cur_n * 64, INS_THREADS), 16384) : blocks_for(cur_n, INS_THREADS)), dim3(INS_THREADS), 0, s,
                           m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, cur_n, walk, kout, ins_wpo, INS_FROM_PREV, (unsigned long long*)nullptr);
    } else {
      hipLaunchKernelGGL(k_insert_keys, dim3(ins_wpo ? std::min<uint32_t>(blocks_for((uint64_t)cur_n * 64, INS_THREADS), 16384) : blocks_for(cur_n, INS_THREADS)), dim3(INS_THREADS), 0, s,
                         m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, cur_n, kin, kout, ins_wpo, 0u, (unsigned long long*)nullptr);
      cur_buf ^= 1u;
    }
    // prep over the survivors: no list, x and y are the high and the low word of the packed keys
    hipLaunchKernelGGL(k_prep, dim3(std::min<uint32_t>(blocks_for(cur_n, PREP_THREADS), m->prep_blocks)), dim3(PREP_THREADS), 0, s,
                       m->d_ctl, m->d_dir, m->dir_size - 1, dir_limit, m->arena.base,
                       (uint64_t)(m->arena.mapped / UNIT_BYTES), (const uint32_t*)nullptr, reinterpret_cast<const uint32_t*>(kout) + 1,
                       reinterpret_cast<const uint32_t*>(kout), m->tasks.p, m->klist.p, m->klist_cap, m->rebal.p, m->fl, 2u, 2u, 0u, (uint2*)nullptr, 0u, (uint32_t*)nullptr);
    HIP_OK(hipGetLastError());
    ctl_read(m, s);
    m->st.rounds++;
    const Ctl& c = *m->h_ctl;
    if (m->trace_rounds) {
      static thread_local double t_prev = 0;
      struct timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      const double now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
      fprintf(stderr, "[smatrix] batch %llu cold round %u (+%.3f ms): keys=%u deferred=%u grow=%u (%llu units) rebal=%u dir_full=%u rows=%u\n",
              (unsigned long long)m->st.batches, round, t_prev ? now - t_prev : 0.0, cur_n, c.n_defer, c.n_tasks, (unsigned long long)c.grow_units,
              c.n_rebal, c.dir_full, c.dir_used);
      t_prev = now;
    }
    if (c.arena_oom) smx_die("internal: arena reservation too small");
    const uint32_t nd = c.n_defer;
    if (nd == 0) break;
    // clustered tables found out HERE (the rule of run_write's rounds): the growth rounds that follow move their big rows in two
    // passes, and the batches that follow start with the hint table and the at-home bitmaps in place: first two steps of the dense-id
    // stream 139 + 37 -> 90 + 15 ms with the wave per key below (what is left of the first: the hot rows' new keys queue at one front per run)
    if (!m->clustered_forced && !m->clustered && (uint64_t)c.n_long_ops * 64 >= cur_n) {
      m->clustered = true; m->clustered_quiet = 0;
      clustered_sync(m, s);
      if (m->trace_rounds) fprintf(stderr, "[smatrix]   %u of %u keys needed the wave-cooperative probe: clustered tables from here on\n", c.n_long_ops, cur_n);
    }
    const bool progress = nd < cur_n || c.n_tasks || c.n_rebal || c.dir_full || m->dir_used != rows_before || (uint64_t)m->dir_used * 2 >= m->dir_size;
    stalled = progress ? 0 : stalled + 1;
    rows_before = m->dir_used;
    m->st.deferred_ops += nd;
    if (c.n_tasks) grow_rows(m, s, c.n_tasks, c.grow_units, c.n_kind, false);
    if (c.n_rebal) {
      hipLaunchKernelGGL(k_rebal, dim3(std::min<uint32_t>(blocks_for(c.n_rebal, 64), 1024)), dim3(64), 0, s, m->d_ctl, m->rebal.p, m->d_dir, m->arena.base);
      HIP_OK(hipGetLastError());
      m->st.rows_rebalanced += c.n_rebal;
    }
    if (c.dir_full) grow_directory(m, 4, s);
    else if ((uint64_t)m->dir_used * 2 >= m->dir_size) grow_directory(m, 2, s);
    kin = kout;
    cur_n = nd;
  }
  // (its scratch -- 325 MB for the first batch of config 2 -- stays until the table has left its youth: the next batches try the
  //  same reduction again, and giving 64-256 MB back to the driver costs 0.2 ms a piece; run_write releases it)
  return true;
}

void run_write(Matrix* m, int op, uint32_t n, const uint32_t* x, const uint32_t* y,
               const uint32_t* v, uint32_t* out, hipStream_t s) {
  if (n == 0) return;
  if (!m->in_cache_sync) m->dirty = true;
  m->defer[0].need(n);
  m->defer[1].need(n);
  if (op == OP_SET) m->cellp.need((size_t)n + AGG_TILE);
  m->set_entries = 0;
  m->st.batches++;
  if (m->dbg_after && m->st.batches == m->dbg_after) {     // measurement builds (SMX_AGG_DBG): switch the kernel's debug mode on
    const uint32_t one = 1;
    HIP_OK(hipMemcpyAsync(&m->d_ctl->pad1, &one, 4, hipMemcpyHostToDevice, s));
  }

  uint32_t cur_n = n;
  const uint32_t* idx = nullptr;
  bool timed0 = m->profile;
  uint32_t stalled = 0, rows_before = m->dir_used;
  m->long_probes = false;
  m->retry_may_repeat = false;
  uint32_t rounds_this_batch = 0;
  bool cold_tried = false;
  bool structure_stable = false;        // round 0 completed the batch: no row was created or doubled (set: the fold's cell addresses hold)
  // the chain is tried when the previous write batch was finished by its round 1 (or by the chain itself)
  // ... and deferred something in its round 0: on a table that takes a batch without a single deferred op -- every key present,
  // or inserted with room to spare -- the chain's dozen launches run empty (65-70 us of a 1.8 ms all-hit step, round 4 trace);
  // such batches go op kernel, prep, read-back, and the first one that defers again is finished by the host-driven loop
  bool chain = m->spec_enabled && m->spec_ready && !m->expect_bulk && n >= m->agg_min && m->dbg_after == 0 &&
               (m->last_nd0 != 0 || m->spec_tiny);
  for (uint32_t round = 0;; round++) {
    // the loop ends when nothing is deferred; it is abandoned only when rounds stop making PROGRESS (a fixed cap
    // would turn a slow but legal batch -- many new rows contending for one directory slot -- into an abort)
    if (stalled > 8) smx_die("write batch did not converge (corrupt row table?)");
    const uint32_t dir_limit = m->dir_size / 2;
    const uint32_t room = dir_limit > m->dir_used ? dir_limit - m->dir_used : 0;
    // at most one growth task / re-partition per row, and only rows named by a deferred op
    m->tasks.need(std::min<uint64_t>(cur_n, m->dir_size));
    m->klist_cap = (uint32_t)std::min<uint64_t>(cur_n, m->dir_size);
    m->klist.need(4 * (size_t)m->klist_cap);
    m->rebal.need(std::min<uint64_t>(cur_n, m->dir_size));
    if (round >= 1 && idx && !cold_tried && m->cold_min && cur_n >= m->cold_min && (uint64_t)cur_n * m->cold_share >= n &&
        op != OP_GET) {
      // a large remainder after the first rounds: the cold start of hot rows (insert_pending_keys); afterwards this round
      // runs over a table in which the keys of `idx` exist
      cold_tried = true;
      if (insert_pending_keys(m, idx, cur_n, x, y, s)) rounds_this_batch += 4;     // (not the steady shape: no chain for the next batch)
    }
    uint32_t* dl = m->defer[round & 1].p;
    const bool chained = chain && round == 0;
    // estimates for the chain's growth round: four times what the previous batch needed (k_grow_plan refuses the rest)
    // (SMATRIX_SPEC_TINY=1, tests: estimates far too small, so that k_grow_plan's refusals and the hand-over to the host-driven
    //  loop are exercised on every chained batch)
    const uint32_t est_nt = m->spec_tiny ? 5u : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * m->spec_nt_prev, 1u << 16), std::min<uint64_t>(cur_n, m->dir_size));
    const uint64_t est_gu = m->spec_tiny ? 24u : std::max<uint64_t>(2 * m->spec_gu_prev, 1ull << 20);
    ensure_arena_free(m, std::min<uint64_t>(cur_n, room) + (chained ? est_gu : 0), s);
    ctl_reset_round(m, s);
    // a large write batch into an EMPTY matrix: no row exists, so round 0 of the op kernel would defer every single op
    // (0.45 ms per 2^24 ops to find that out): the deferred list is the batch itself, in order, and the bulk path takes over
    // (incr / decr only: a set batch's round 0 is k_set_fold, whose entries the passes after the rounds need)
    const bool all_new = round == 0 && !chained && m->dir_used == 0 && m->bulk_enabled && m->expect_bulk && n >= m->fix_min &&
                         n >= m->agg_min && m->dbg_after == 0 && (op == OP_INCR || op == OP_DECR);
    if (all_new) {
      hipLaunchKernelGGL(k_iota, dim3(std::min<uint32_t>(blocks_for(n), 4096)), dim3(256), 0, s, dl, n);
      HIP_OK(hipGetLastError());
      HIP_OK(hipMemcpyAsync(&m->d_ctl->n_defer, &n, 4, hipMemcpyHostToDevice, s));
      timed0 = false;
    } else {
      // (round 5) when the pass in front of prep follows this op round (the condition of `pre_pass` below), the clustered folding
      // kernel sets the ops that wait for prep aside in that pass's OUTPUT list -- keys known to be absent from rows at their
      // threshold: 30 % of the deferred ops of a late dense-id batch, 80-90 % of a young table's, each of which cost the pass a
      // wave's trip for nothing (ArenaHead::absent_list, Ctl::n_absent)
      if (op == OP_INCR || op == OP_DECR) {
        const bool pass_follows = (chained || (round == 0 && idx == nullptr && m->home_on && m->far_join && n >= (1u << 16))) && m->clustered;
        absent_list_set(m, pass_follows && m->absent_split && m->dbg_after == 0 ? m->defer[1].p : nullptr, s);
      }
      launch_apply_op(m, op, s, cur_n, idx, x, y, v, out, dl);
    }
    DBG_STEP(m, s, m->long_probes ? "op kernel (lane per op)" : "op kernel");
#if defined(SMX_AGG_DBG) && SMX_AGG_DBG == 5
    // measurement build "inserts without tickets": rows overfill and their deferred ops never converge -- only the
    // round-0 launch is of interest, what it deferred is dropped
    if (m->dbg_after && m->st.batches >= m->dbg_after) {
      ctl_read(m, s);
      if (timed0) { account_kernel_time(m, op, n); timed0 = false; }
      break;
    }
#endif
    if (round == 0 && !chained && m->bulk_enabled && m->expect_bulk && op != OP_GET && n >= m->fix_min) {
      // the previous batch deferred a large share of its ops (bulk load, young matrix): look at this one's count
      // before prep -- one extra read-back, only in this regime -- and group a large remainder by row instead of
      // walking it through a round per doubling
      if (!all_new) ctl_read(m, s);                          // (an empty matrix defers the whole batch: nothing to read)
      if (timed0) { account_kernel_time(m, op, n); timed0 = false; }
      const uint32_t nd0 = all_new ? n : m->h_ctl->n_defer + m->h_ctl->n_absent;
      m->expect_bulk = (uint64_t)nd0 * 8 >= n;
      // worth it when a LARGE share of the batch is pending (bulk loads, the first batch of a matrix: every op names a
      // row that does not exist yet).  At 10-15 % -- batches 1 and 2 of config 2 -- grouping costs more than the rounds
      // it replaces (round 3, same box: step 1 7.2 -> 3.9 ms, step 2 5.0 -> 3.2 ms without it): SMATRIX_BULK_SHARE
      if (nd0 >= m->fix_min && nd0 < 0x80000000u && (uint64_t)nd0 * m->fix_share >= n) {   // (k_fix_scatter keeps a flag in bit 31 of a position)
        m->st.rounds++;
        rounds_this_batch++;
        m->st.deferred_ops += nd0;
        m->tasks.need(std::min<uint64_t>(cur_n, m->dir_size));
        if (!all_new && m->h_ctl->n_absent) {
          // (the folding kernel kept two lists for a pass that does not follow now: one list again)
          HIP_OK(hipMemcpyAsync(dl + m->h_ctl->n_defer, m->defer[1].p, (size_t)m->h_ctl->n_absent * 4, hipMemcpyDeviceToDevice, s));
          HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&m->d_ctl->n_defer), (int)nd0, 1, s));
          HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&m->d_ctl->n_absent), 0, 1, s));
        }
        uint32_t* dl2 = m->defer[1].p;
        const uint32_t nd2 = op == OP_INCR   ? run_bulk_t<OP_INCR>(m, nd0, dl, dl2, x, y, v, out, s)
                             : op == OP_DECR ? run_bulk_t<OP_DECR>(m, nd0, dl, dl2, x, y, v, out, s)
                                             : run_bulk_t<OP_SET>(m, nd0, dl, dl2, x, y, v, out, s);
        if (m->trace_rounds)
          fprintf(stderr, "[smatrix] batch %llu bulk path: %u deferred ops grouped by row, %u handed back, rows=%u\n",
                  (unsigned long long)m->st.batches, nd0, nd2, m->dir_used);
        if (nd2 == 0) break;
        m->retry_may_repeat = true;      // (the batch's own ops: y = 0 ops, the hot rows' surplus)
        idx = dl2;                       // what was handed back sits in defer[1]: the next round must write defer[0],
        cur_n = nd2;                     // so it is numbered 2
        round++;
        rounds_this_batch += 2;          // (not the steady shape: no chain for the next batch)
        continue;
      }
    }
    const auto launch_prep = [&](const uint32_t* list) {
      // (a short list of a clustered matrix is taken a wave per op: the grid covers 64 lanes per op then)
      const uint32_t pgrid = m->clustered && cur_n <= 8192u ? blocks_for((uint64_t)cur_n * 64, PREP_THREADS) : blocks_for(cur_n, PREP_THREADS);
      // (clustered matrices: the ops found absent leave their keys for the growth round -- growth.hpp, k_pend_group)
      const bool pend = m->clustered && m->home_on && m->pend_on && m->in_stride != 3;
      if (pend) {
        m->pend_rec.need(std::min<size_t>(n, (size_t)1 << 22));
        m->pend_ctl.need(16);
        HIP_OK(hipMemsetAsync(m->pend_ctl.p, 0, 16, s));
        m->pend_est = (uint32_t)std::min<uint64_t>(chained ? std::max<uint64_t>(2ull * m->spec_nd_prev, 1u << 16) : cur_n, m->pend_rec.cap);
      }
      m->pend_armed = pend;
      hipLaunchKernelGGL(k_prep, dim3(std::min<uint32_t>(pgrid, m->prep_blocks)), dim3(PREP_THREADS), 0, s,
                         m->d_ctl, m->d_dir, m->dir_size - 1, dir_limit, m->arena.base,
                         (uint64_t)(m->arena.mapped / UNIT_BYTES), list, x, y, m->tasks.p, m->klist.p, m->klist_cap,
                         m->rebal.p, m->fl, m->in_stride, 0u, m->clustered ? 8192u : 0u, pend ? m->pend_rec.p : nullptr, (uint32_t)m->pend_rec.cap, m->pend_ctl.p);
    };
    // Clustered tables (dense ids): the folding kernel sets every op aside whose probe outruns its budget -- 770 000 of a
    // 2^24-op batch of the dense stream, nearly all of them HITS on keys that sit far from home -- and prep then walked
    // each of those probes to the end only to find the key present.  In the chained shape the whole deferred list takes
    // a wave-per-op pass first (k_apply_wpo); prep, the growth round and the retry see what that pass leaves: the ops
    // that really wait for a structure change.
    // (round 5) with the far join the pass also pays in front of the FIRST prep of a batch the host drives round by round -- the
    // young table's batches, whose deferred lists are the longest
    const bool pre_pass = (chained || (round == 0 && idx == nullptr && !all_new && m->home_on && m->far_join && n >= (1u << 16))) && m->clustered &&
                          (op == OP_INCR || op == OP_DECR);
    bool far_joined = false;
    if (pre_pass) {
      uint32_t* dlp = m->defer[1].p;
      hipLaunchKernelGGL(k_round_advance, dim3(1), dim3(64), 0, s, m->d_ctl, m->rebal.p, m->d_dir, m->arena.base);
      // (the table is sized from the list the last join saw -- the folding kernel's deferred ops, 2-4x what the pass leaves)
      // (a round the host drives anyway: the list's length is read instead -- the first joins of a young table, whose lists are the
      //  longest, ran with a table sized for 2^16 keys and overflowed: 9 ms for the pass of the dense stream's third batch)
      uint64_t est_far = std::max<uint64_t>({2ull * m->spec_nd_prev, 3ull * m->far_nd_seen / 2, 1ull << 16});
      if (!chained && m->far_join && m->home_on) {
        uint32_t nd_now = 0;
        HIP_OK(hipMemcpyAsync(&nd_now, &m->d_ctl->n_prev, 4, hipMemcpyDeviceToHost, s));      // (k_round_advance has just moved it there)
        sync_counted(m, s);
        est_far = (uint64_t)nd_now + nd_now / 8 + 1024;
        if (m->trace_rounds) fprintf(stderr, "[smatrix] batch %llu: the folding kernel deferred %u ops\n", (unsigned long long)m->st.batches, nd_now);
      }
      far_joined = far_join_enqueue(m, s, dl, x, y, (uint32_t)std::min<uint64_t>(est_far, cur_n));
      if (far_joined) {
        const dim3 wgrid(65536);
        if (op == OP_INCR) hipLaunchKernelGGL((k_apply_wpo_far<OP_INCR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dlp, m->in_stride);
        else hipLaunchKernelGGL((k_apply_wpo_far<OP_DECR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dlp, m->in_stride);
      } else {
      const dim3 wgrid(65536);
      if (op == OP_INCR) hipLaunchKernelGGL((k_apply_wpo<OP_INCR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dlp, m->in_stride);
      else hipLaunchKernelGGL((k_apply_wpo<OP_DECR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dlp, m->in_stride);
      }
      HIP_OK(hipGetLastError());
      DBG_STEP(m, s, "the pass in front of prep");
      dl = dlp;
    }
    launch_prep(dl);
    HIP_OK(hipGetLastError());
    if (far_joined) arena_head_set(m, offsetof(ArenaHead, far_on), 0u, s);     // (rows are about to double: the join's view of the tables ends here)
    DBG_STEP(m, s, "k_prep");
    uint32_t nd_chain0 = 0;
    bool retry_halves = false;
    if (chained) {
      // ---- the rest of the chain: growth for the rows round 0's prep flagged, the retry, its prep -- no read-back between
      const uint32_t est_nk[4] = {std::max<uint32_t>(2 * m->spec_nk_prev[0], 4096), std::max<uint32_t>(2 * m->spec_nk_prev[1], 1024),
                                  std::max<uint32_t>(2 * m->spec_nk_prev[2], 256), std::max<uint32_t>(2 * m->spec_nk_prev[3], 64)};
      const uint64_t host_next = m->arena_next;
      grow_rows(m, s, est_nt, est_gu, est_nk, true);
      m->arena_next = host_next;                              // (the mirror is refreshed by the read-back below)
      hipLaunchKernelGGL(k_round_advance, dim3(1), dim3(64), 0, s, m->d_ctl, m->rebal.p, m->d_dir, m->arena.base);
      // the retry: lane per op over the device-side list (its length is ctl->n_prev), grid for 4x the previous batch's
      const uint32_t est_nd = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * m->spec_nd_prev, 1u << 16), cur_n);
      uint32_t* dl1 = pre_pass ? m->defer[0].p : m->defer[1].p;       // (the list the retry reads sits in the other one)
      const dim3 rgrid(std::min<uint32_t>(blocks_for(est_nd), 16384));
      if (m->clustered && est_nd <= m->wpo_max && m->retry_split && m->d_hints) {
        // the retry in two halves (k_apply_short): a lane per op for what is short again, then a wave per op over the rest; the
        // list that is left ends up in the buffer the retry READ, so the round's parity moves on by one (below)
        uint32_t* dl0 = dl;
        const dim3 wgrid(std::min<uint32_t>(blocks_for((uint64_t)est_nd * (64 / SMX_WPO_OPS)), 65536));
        switch (op) {
          case OP_SET:  hipLaunchKernelGGL((k_apply_short<OP_SET>), rgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
          case OP_INCR: hipLaunchKernelGGL((k_apply_short<OP_INCR>), rgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
          default:      hipLaunchKernelGGL((k_apply_short<OP_DECR>), rgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
        }
        hipLaunchKernelGGL(k_list_advance, dim3(1), dim3(64), 0, s, m->d_ctl);
        switch (op) {
          case OP_SET:  hipLaunchKernelGGL((k_apply_wpo<OP_SET>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl1, x, y, v, out, dl0, m->in_stride); break;
          case OP_INCR: hipLaunchKernelGGL((k_apply_wpo<OP_INCR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl1, x, y, v, out, dl0, m->in_stride); break;
          default:      hipLaunchKernelGGL((k_apply_wpo<OP_DECR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl1, x, y, v, out, dl0, m->in_stride); break;
        }
        dl1 = dl0;
        retry_halves = true;
      } else if (m->clustered && est_nd <= m->wpo_max) {
        const dim3 wgrid(std::min<uint32_t>(blocks_for((uint64_t)est_nd * 64), 65536));      // (a wave per op: launch_apply)
        switch (op) {
          case OP_SET:  hipLaunchKernelGGL((k_apply_wpo<OP_SET>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
          case OP_INCR: hipLaunchKernelGGL((k_apply_wpo<OP_INCR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
          default:      hipLaunchKernelGGL((k_apply_wpo<OP_DECR>), wgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
        }
      } else
      switch (op) {
        case OP_SET:  hipLaunchKernelGGL((k_apply<OP_SET>), rgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
        case OP_INCR: hipLaunchKernelGGL((k_apply<OP_INCR>), rgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
        default:      hipLaunchKernelGGL((k_apply<OP_DECR>), rgrid, dim3(256), 0, s, m->d_ctl, m->d_dir, m->dir_size - 1, m->arena.base, 0xFFFFFFFFu, dl, x, y, v, out, dl1, m->in_stride); break;
      }
      launch_prep(dl1);
      HIP_OK(hipGetLastError());
      dl = dl1;
    }
    ctl_read(m, s);
    if (timed0 && round == 0) account_kernel_time(m, op, n);
    m->st.rounds++;
    rounds_this_batch++;
    if (chained) {
      // bookkeeping of the round the host did not see, then on as if round 1 had just been read back
      const Ctl& c = *m->h_ctl;
      nd_chain0 = c.spec_nd0;
      if (far_joined) { m->far_rows_seen = c.n_big; m->far_units_seen = c.n_units; m->far_nd_seen = c.far_nd; }
      if (!m->in_cache_sync) m->last_nd0 = nd_chain0;
      m->expect_bulk = (uint64_t)nd_chain0 * 8 >= n;
      m->st.spec_chains++;
      if (nd_chain0) {
        m->st.rounds++;
        rounds_this_batch++;
        m->st.deferred_ops += nd_chain0;
        m->st.rows_grown += c.spec_nt0;                        // (refused tasks are counted again when they do grow: rare)
        m->st.rows_rebalanced += c.spec_nrebal0;
        m->spec_nd_prev = nd_chain0; m->spec_nt_prev = c.spec_nt0; m->spec_gu_prev = c.spec_gu0;
        for (int k = 0; k < 4; k++) m->spec_nk_prev[k] = c.spec_nkind0[k];
      }
      if (c.spec_failed) {
        m->st.spec_refused++;
        const uint32_t zero = 0;
        HIP_OK(hipMemcpyAsync(&m->d_ctl->spec_failed, &zero, 4, hipMemcpyHostToDevice, s));
      }
      if (m->trace_rounds && m->rest_dbg) {
        static thread_local unsigned long long last[3] = {0, 0, 0};
        unsigned long long now[3];
        HIP_OK(hipMemcpy(now, m->rest_dbg + 16, 24, hipMemcpyDeviceToHost));
        fprintf(stderr, "[smatrix]   far join: big rows %u (%u units); long probes of the pass: not in the table %llu, cell known %llu, absent %llu\n",
                c.n_big, c.n_units, now[0] - last[0], now[1] - last[1], now[2] - last[2]);
        memcpy(last, now, sizeof last);
      }
      if (m->trace_rounds)
        fprintf(stderr, "[smatrix] batch %llu chain: ops=%u (the pass in front of prep took %u) deferred=%u grow=%u (%llu units) rebal=%u refused=%u | after the retry: deferred=%u grow=%u rows=%u | long probes %u%s\n",
                (unsigned long long)m->st.batches, cur_n, far_joined ? c.far_nd : 0u, nd_chain0, c.spec_nt0, (unsigned long long)c.spec_gu0, c.spec_nrebal0,
                c.spec_failed, c.n_defer, c.n_tasks, c.dir_used, c.n_long_ops, m->clustered ? " (clustered)" : "");
      // clustered mode goes off again after 8 chained batches in a row with hardly a long probe (the ids have changed their
      // nature: the wave-per-op pass in front of prep costs a scrambled-id batch 1.2 ms)
      if (!m->clustered_forced && m->clustered) {
        // (with a hint table the folding kernel finishes most far hits itself and counts (tile, key) ENTRIES, one in 256: a
        //  hot far key is one entry per tile, not thousands of ops -- the bar is lower by that much)
        m->clustered_quiet = (uint64_t)c.n_long_ops * (m->d_hints ? 4096 : 256) < n ? m->clustered_quiet + 1 : 0;
        if (m->clustered_quiet >= 8) { m->clustered = false; clustered_sync(m, s); }
      }
      if (nd_chain0 == 0) { structure_stable = true; break; }  // round 0 deferred nothing: the rest of the chain ran empty
      cur_n = nd_chain0;
      round = (pre_pass ? 2 : 1) + (retry_halves ? 1 : 0);     // (the next round writes the list that `dl` is NOT)
    } else if (pre_pass) {
      // the host-driven round 0 with the pass in front of prep: the list that is left sits in defer[1], so the next round is numbered 2
      if (far_joined) { m->far_rows_seen = m->h_ctl->n_big; m->far_units_seen = m->h_ctl->n_units; m->far_nd_seen = m->h_ctl->far_nd; }
      if (m->trace_rounds) {
        uint32_t ovf = 0;
        HIP_OK(hipMemcpy(&ovf, m->arena.base + offsetof(ArenaHead, far_overflow), 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[smatrix] batch %llu round 0 with the pass in front of prep (it took %u ops; join table 2^%u%s): ops=%u deferred=%u grow=%u rows=%u\n", (unsigned long long)m->st.batches,
                far_joined ? m->h_ctl->far_nd : 0u, m->far_tab_lg, ovf ? ", OVERFLOWED" : "", cur_n, m->h_ctl->n_defer, m->h_ctl->n_tasks, m->h_ctl->dir_used);
      }
    } else if (m->trace_rounds) {
      static thread_local double t_prev = 0;
      struct timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      const double now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
      fprintf(stderr, "[smatrix] batch %llu round %u (+%.3f ms): ops=%u deferred=%u grow=%u (%llu units) rebal=%u dir_full=%u rows=%u\n",
              (unsigned long long)m->st.batches, round, t_prev ? now - t_prev : 0.0, cur_n, m->h_ctl->n_defer,
              m->h_ctl->n_tasks, (unsigned long long)m->h_ctl->grow_units, m->h_ctl->n_rebal, m->h_ctl->dir_full,
              m->h_ctl->dir_used);
      t_prev = now;
    }
    if (m->h_ctl->arena_oom) smx_die("internal: arena reservation too small");
    const uint32_t nd = m->h_ctl->n_defer;
    if (round == 0) m->expect_bulk = (uint64_t)nd * 8 >= n;
    if (round == 0 && !chained && !m->in_cache_sync) m->last_nd0 = nd;
    if (round == 1 && !chained && nd == 0) {
      // the steady shape: remember what round 0 needed -- the next batch's chain is sized from it
      m->spec_nd_prev = cur_n;
    }
    if (nd == 0) { structure_stable = round == 0 && !chained; break; }
    if (m->h_ctl->n_long) { m->long_probes = true; m->st.long_probe_rounds++; }
    // clustered: percents of a batch needed the wave-cooperative probe (dense ids: 4-5 %; any large table at load 1/2 has a few
    // sequences beyond the budget -- the first batches of the scrambled stream do -- and must not switch it on)
    if (!m->clustered_forced && !m->clustered && (uint64_t)m->h_ctl->n_long_ops * 64 >= n) { m->clustered = true; m->clustered_quiet = 0; }
    clustered_sync(m, s);

    const bool progress = nd < cur_n || m->h_ctl->n_long || m->h_ctl->n_tasks || m->h_ctl->n_rebal || m->h_ctl->dir_full ||
                          m->dir_used != rows_before || (uint64_t)m->dir_used * 2 >= m->dir_size;
    stalled = progress ? 0 : stalled + 1;
    m->retry_may_repeat = m->dir_used != rows_before;        // ops of rows that did not exist were deferred whatever their key
    rows_before = m->dir_used;
    m->st.deferred_ops += nd;
    if (round == 0) {
      m->spec_nt_prev = m->h_ctl->n_tasks; m->spec_gu_prev = m->h_ctl->grow_units;
      for (int k = 0; k < 4; k++) m->spec_nk_prev[k] = m->h_ctl->n_kind[k];
    }
    if (m->h_ctl->n_tasks) grow_rows(m, s, m->h_ctl->n_tasks, m->h_ctl->grow_units, m->h_ctl->n_kind, false);
    if (m->h_ctl->n_rebal) {
      hipLaunchKernelGGL(k_rebal, dim3(std::min<uint32_t>(blocks_for(m->h_ctl->n_rebal, 64), 1024)), dim3(64), 0, s,
                         m->d_ctl, m->rebal.p, m->d_dir, m->arena.base);
      HIP_OK(hipGetLastError());
      m->st.rows_rebalanced += m->h_ctl->n_rebal;
    }
    if (m->h_ctl->dir_full) grow_directory(m, 4, s);
    else if ((uint64_t)m->dir_used * 2 >= m->dir_size) grow_directory(m, 2, s);
    idx = dl;
    cur_n = nd;
    if (pre_pass && !chained) round++;                       // (what is left sits in defer[1]: the next round writes defer[0])
  }
  // the chain is for batches near the steady shape (a few rounds: whatever its rounds 0 and 1 leave is finished by the
  // host-driven loop at no extra cost); young tables with many rounds per batch stay host-driven
  m->spec_ready = rounds_this_batch <= 4;
  for (void* q : m->put_aside) (void)dev_free(q);
  m->put_aside.clear();
  // the cold start's scratch (and the key set the bulk path's row count may have left behind) goes back once a batch has
  // had the steady shape -- or at once when this batch made no use of it beyond the row count
  if (m->spec_ready || !cold_tried) {
    m->cold_set.release_on(s); m->cold_keys[0].release_on(s); m->cold_keys[1].release_on(s); m->cold_keys[2].release_on(s);
    m->cold_idx[0].release_on(s); m->cold_idx[1].release_on(s); m->cold_zero.release_on(s);
  }
  if (m->trace_rounds) {
    AllocClock& ac = alloc_clock();
    if (ac.n_alloc + ac.n_free)
      fprintf(stderr, "[smatrix] batch %llu: %llu device allocations %.3f ms, %llu frees %.3f ms\n", (unsigned long long)m->st.batches,
              (unsigned long long)ac.n_alloc, ac.s_alloc * 1e3, (unsigned long long)ac.n_free, ac.s_free * 1e3);
    ac = AllocClock();
  }

  if (op == OP_SET && m->set_entries) {
    // highest-index-wins across tiles, over the winners of k_set_fold only (locate also clears the value word)
    const uint32_t ne = m->set_entries;
    dim3 g(blocks_for(ne)), b(256);
    // (round 0 completed the batch: every entry's cell is where k_set_fold found or created it, already cleared)
    if (!structure_stable || m->set_always_locate)
      hipLaunchKernelGGL(k_set_locate_e, g, b, 0, s, m->d_dir, m->dir_size - 1, m->arena.base, ne, x, y, m->ent_idx.p, m->cellp.p, m->in_stride);
    else m->st.set_located_by_fold++;
    hipLaunchKernelGGL(k_set_rank_e, g, b, 0, s, ne, m->ent_idx.p, m->cellp.p, m->arena.base);
    hipLaunchKernelGGL(k_set_pick_e, g, b, 0, s, ne, m->ent_idx.p, m->cellp.p, m->arena.base);
    hipLaunchKernelGGL(k_set_store_e, g, b, 0, s, ne, m->ent_idx.p, m->cellp.p, v, m->arena.base, m->in_stride);
    HIP_OK(hipGetLastError());
    sync_counted(m, s);
  } else if (op == OP_SET) {
    dim3 g(blocks_for(n)), b(256);
    hipLaunchKernelGGL(k_set_locate, g, b, 0, s, m->d_dir, m->dir_size - 1, m->arena.base, n, x, y,
                       m->cellp.p, m->in_stride);
    hipLaunchKernelGGL(k_set_clear, g, b, 0, s, n, m->cellp.p, m->arena.base);
    hipLaunchKernelGGL(k_set_rank, g, b, 0, s, n, m->cellp.p, m->arena.base);
    hipLaunchKernelGGL(k_set_pick, g, b, 0, s, n, m->cellp.p, m->arena.base);
    hipLaunchKernelGGL(k_set_store, g, b, 0, s, n, m->cellp.p, v, m->arena.base, m->in_stride);
    HIP_OK(hipGetLastError());
    sync_counted(m, s);
  }
}

void run_get(Matrix* m, uint32_t n, const uint32_t* x, const uint32_t* y, uint32_t* out,
             hipStream_t s) {
  if (n == 0) return;
  if (!m->profile) {
    launch_apply_op(m, OP_GET, s, n, nullptr, x, y, nullptr, out, nullptr);
    return;
  }
  get_timing_resolve(m, 15);
  Matrix::TimedLaunch t{get_timing_event(m), get_timing_event(m), n};
  HIP_OK(hipEventRecord(t.e0, s));
  m->profile = false;                                   // (launch_apply_op would record the shared pair of the write path)
  launch_apply_op(m, OP_GET, s, n, nullptr, x, y, nullptr, out, nullptr);
  m->profile = true;
  HIP_OK(hipEventRecord(t.e1, s));
  m->get_pending.push_back(t);
}

Matrix* M(smatrix_t* self) { return static_cast<Matrix*>(self->impl); }

void refresh_public(smatrix_t* self) {
  Matrix* m = M(self);
  uint64_t free_units = 0;
  for (uint32_t c = 0; c < N_CLASSES; c++)
    if (m->free_cnt[c] > 0) free_units += (uint64_t)m->free_cnt[c] * block_units(c + ROW_FIRST_LG);
  const uint64_t live_units = m->arena_next > free_units ? m->arena_next - free_units : 0;
  self->mem = (uint64_t)m->dir_size * sizeof(DirSlot) + live_units * UNIT_BYTES;
}

bool file_flush(smatrix_t* self, Matrix* m, bool all, std::unique_lock<std::mutex>* unlock);
void cache_sync(Matrix* m, bool drop);

void apply_dev_locked(smatrix_t* self, int op, size_t n, const uint32_t* x, const uint32_t* y,
                      const uint32_t* v, uint32_t* out, hipStream_t s) {
  Matrix* m = M(self);
  if (n >= (1ull << 32)) smx_die("batch too large (n must be < 2^32)");
  if (op == OP_GET) run_get(m, (uint32_t)n, x, y, out, s);
  else {
    // (what the call costs the host: smatrix_stats_t::write_call_ms / write_wait_ms / write_alloc_ms)
    const double t0 = mono_s(), w0 = m->w_wait_s;
    const AllocClock a0 = alloc_clock();
    run_write(m, op, (uint32_t)n, x, y, v, out, s);
    const AllocClock a1 = alloc_clock();
    const double call = mono_s() - t0, wait = m->w_wait_s - w0, al = std::max(0.0, (a1.s_alloc + a1.s_free) - (a0.s_alloc + a0.s_free));
    m->w_call_s += call; m->w_alloc_s += al;
    m->st.write_call_ms = m->w_call_s * 1e3; m->st.write_wait_ms = m->w_wait_s * 1e3; m->st.write_alloc_ms = m->w_alloc_s * 1e3;
    m->st.last_write_call_ms = call * 1e3; m->st.last_write_wait_ms = wait * 1e3; m->st.last_write_alloc_ms = al * 1e3;
  }
  refresh_public(self);
  // SMATRIX_FLUSH_EVERY=N: the backing file is brought up to date after every N-th write batch (the reference's IO
  // thread writes dirty rows behind the caller's back all the time, src/smatrix.c:929-960; here it is a checkpoint)
  // (round 6: the checkpoint itself is taken by the public entry point once it has let go of m->mu -- CkptAfter, checkpoint():
  //  it needs file_mu, which is never waited for with the matrix lock held; a host-pointer call in chunks owes ONE)
  if (op != OP_GET && m->flush_every && self->fd && !m->in_cache_sync && m->st.batches % m->flush_every == 0) m->ckpt_due = true;
}

// Brings the file up to date (SMATRIX_FLUSH_EVERY).  The protocol of smatrix_flush -- ADVICE r3: a plain `dirty = false` lost
// the flag of a lock-free scalar write that had landed on another mirrored cell a moment before, and smatrix_close then
// skipped its final flush.  Called WITHOUT m->mu.
void checkpoint(smatrix_t* self) {
  Matrix* m = M(self);
  set_device(m);
  std::unique_lock<std::mutex> fg(m->file_mu);
  std::unique_lock<std::mutex> g(m->mu);
  if (!self->fd || !m->dirty.exchange(false)) return;
  cache_sync(m, false);
  if (file_flush(self, m, false, &g)) m->dirty = true;
}
// Declared in a public entry point BEFORE it locks m->mu: runs after the lock is gone.
struct CkptAfter {
  smatrix_t* self;
  explicit CkptAfter(smatrix_t* s) : self(s) {}
  ~CkptAfter() { if (M(self)->ckpt_due.exchange(false)) checkpoint(self); }
};

// Writes the scalar ABI's mirrored values back (one batched set of cells that all exist: no structure change) before
// anything reads the tables; `drop` additionally forgets the mirror (before writes that may touch mirrored cells).
// Caller holds m->mu.
void cache_sync(Matrix* m, bool drop) {
  if (!m->cache.enabled) return;
  std::vector<uint32_t> xs, ys, vs;
  m->cache.drain(xs, ys, vs, drop);
  const size_t k = xs.size();
  if (!k) return;
  hipStream_t s = m->stream;
  m->sx.need(k); m->sy.need(k); m->sv.need(k); m->so.need(k);
  HIP_OK(hipMemcpyAsync(m->sx.p, xs.data(), k * 4, hipMemcpyHostToDevice, s));
  HIP_OK(hipMemcpyAsync(m->sy.p, ys.data(), k * 4, hipMemcpyHostToDevice, s));
  HIP_OK(hipMemcpyAsync(m->sv.p, vs.data(), k * 4, hipMemcpyHostToDevice, s));
  const uint32_t keep = m->in_stride;
  m->in_stride = 1;
  const uint64_t batches = m->st.batches;
  // the write-back is not one of the caller's batches: the heuristics the next batch is planned from (speculative chain,
  // bulk path, clustered mode) keep what the caller's last batch left (with the flusher calling this every
  // 100 ms a young matrix got the chain and lost the bulk path on its next batch)
  const bool k_ready = m->spec_ready, k_bulk = m->expect_bulk, k_long = m->long_probes, k_clu = m->clustered;
  const uint32_t k_nd = m->spec_nd_prev, k_nt = m->spec_nt_prev, k_quiet = m->clustered_quiet;
  const uint64_t k_gu = m->spec_gu_prev;
  uint32_t k_nk[4];
  memcpy(k_nk, m->spec_nk_prev, sizeof k_nk);
  m->in_cache_sync = true;
  run_write(m, OP_SET, (uint32_t)k, m->sx.p, m->sy.p, m->sv.p, m->so.p, s);   // synchronises (set resolves duplicates last)
  m->in_cache_sync = false;
  m->spec_ready = k_ready; m->expect_bulk = k_bulk; m->long_probes = k_long; m->clustered = k_clu;
  clustered_sync(m, s);
  m->spec_nd_prev = k_nd; m->spec_nt_prev = k_nt; m->clustered_quiet = k_quiet; m->spec_gu_prev = k_gu;
  memcpy(m->spec_nk_prev, k_nk, sizeof k_nk);
  m->st.batches = batches;                                                     // bookkeeping of the caller's batches only
  m->in_stride = keep;
  m->cache.flushes++;
  m->cache.flushed_cells += k;
}

}  // namespace

// ---- persistence (src/smatrix.c:30-72 file format) ---------------------------------
#include "smx_file.inc"

namespace {
void flusher_main(smatrix_t* self, Matrix* m) {
  using clock = std::chrono::steady_clock;
  const auto period = std::chrono::milliseconds(m->flush_ms);
  std::unique_lock<std::mutex> l(m->fl_mu);
  auto next = clock::now() + period;
  while (!m->fl_stop) {
    m->fl_cv.wait_until(l, next, [&] { return m->fl_stop; });
    if (m->fl_stop) break;
    l.unlock();
    auto pause = period;
    if (m->dirty.load()) {
      const auto t0 = clock::now();
      bool more = false;
      {
        set_device(m);
        std::lock_guard<std::mutex> fg(m->file_mu);       // (first: see Matrix::file_mu)
        std::unique_lock<std::mutex> g(m->mu);
        if (m->dirty.exchange(false)) {
          cache_sync(m, false);
          // (round 4) the lock is held while the dirty rows are collected, laid out and snapshot on the device, and goes
          // back to the callers BEFORE the bytes are copied to the host and written (file_flush releases `g`)
          more = file_flush(self, m, false, &g);
          m->file_bg_flushes_done.fetch_add(1);
          if (more) m->dirty = true;                 // the snapshot budget was reached: the rest goes out with the next one
        }
      }
      // the device copies and the writes of a flush compete with the callers for PCIe and the page cache: keep them to
      // about a tenth of the time (small matrices stay at the period); a flush that left rows behind goes on at once
      const auto took = std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - t0);
      pause = more ? std::chrono::milliseconds(1) : std::max(period, took * 10);
    }
    l.lock();
    next = clock::now() + pause;
  }
}
}  // namespace

// ---- large host-pointer batches: staged in chunks, three stages overlapped -------------------------------------------------
// smatrix_apply_batch / smatrix_rowlen_batch are what a batch binding calls with the caller's own arrays (the reference's glue
// passes host values, src/smatrix_jni.c:95-111).  Round 4 copied the whole arrays with three pageable hipMemcpyAsync, ran the
// kernels and copied the results back, one after the other: 1.3 G incr/s and 2.0 G get/s per 2^24-op call, against a PCIe bound
// of 3.4-4.5 G/s -- a pageable copy is the runtime's single-threaded memcpy into its own pinned buffer.  Calls above two chunks
// now run as a three-stage pipeline over chunks of 2^21 ops (SMATRIX_HOST_CHUNK_LG) and three sets of buffers:
//   feeder thread   the caller's arrays -> pinned memory with a pool of copy threads, then host-to-device on its own stream
//   calling thread  the op kernels of chunk k (the matrix lock is the caller's, as before), in order: a write batch's chunks
//                   are applied one after the other, so duplicates resolve exactly as in one call (set: the later op wins)
//   drain thread    device-to-host on a third stream, then pinned memory -> the caller's result array
// so that the transfer of chunk k+1, the kernels of chunk k and the return of chunk k-1 overlap.
struct CopyPool {                       // a few persistent threads that run memcpy slices
  // (bad != nullptr: not a copy -- are all `len` / 4 words at src equal to `want`?  *bad is set when one is not)
  struct Slice { unsigned char* dst; const unsigned char* src; size_t len; std::atomic<uint32_t>* bad = nullptr; uint32_t want = 0; };
  std::vector<std::thread> th;
  std::mutex mu, run_mu;
  std::condition_variable cv, done_cv;
  std::vector<Slice> slices;
  size_t next = 0, left = 0;
  bool stop = false;
  void start(unsigned n) {
    for (unsigned t = 0; t < n; t++)
      th.emplace_back([this] {
        std::unique_lock<std::mutex> l(mu);
        for (;;) {
          cv.wait(l, [&] { return stop || next < slices.size(); });
          if (stop) return;
          const Slice sl = slices[next++];
          l.unlock();
          if (!sl.bad) memcpy(sl.dst, sl.src, sl.len);
          else if (!sl.bad->load(std::memory_order_relaxed)) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(sl.src);
            const size_t nw = sl.len / 4;
            uint32_t diff = 0;
            for (size_t i = 0; i < nw && !diff; i += 1024) {           // (blocks without a branch inside: the compiler vectorises them)
              const size_t e = std::min(nw, i + 1024);
              for (size_t q = i; q < e; q++) diff |= p[q] ^ sl.want;
            }
            if (diff) sl.bad->store(1, std::memory_order_relaxed);
          }
          l.lock();
          if (--left == 0) done_cv.notify_all();
        }
      });
  }
  void add(std::vector<Slice>& v, void* dst, const void* src, size_t len) {
    for (size_t o = 0; o < len; o += (size_t)1 << 20) v.push_back({static_cast<unsigned char*>(dst) + o, static_cast<const unsigned char*>(src) + o, std::min<size_t>((size_t)1 << 20, len - o)});
  }
  void add_check(std::vector<Slice>& v, const void* src, size_t len, uint32_t want, std::atomic<uint32_t>* bad) {
    for (size_t o = 0; o < len; o += (size_t)1 << 20) v.push_back({nullptr, static_cast<const unsigned char*>(src) + o, std::min<size_t>((size_t)1 << 20, len - o), bad, want});
  }
  void run(std::vector<Slice>&& v) {     // returns when every slice is copied
    if (v.empty()) return;
    std::lock_guard<std::mutex> one(run_mu);
    std::unique_lock<std::mutex> l(mu);
    slices = std::move(v); next = 0; left = slices.size();
    cv.notify_all();
    done_cv.wait(l, [&] { return left == 0; });
    slices.clear(); next = 0;
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> l(mu); stop = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};

struct HostPipe {
  static constexpr int NB = 3;
  size_t chunk = 0;                      // ops per chunk
  uint32_t* h_in[NB] = {nullptr, nullptr, nullptr};    // pinned: up to three arrays of `chunk` words
  uint32_t* h_out[NB] = {nullptr, nullptr, nullptr};
  uint32_t* d_in[NB] = {nullptr, nullptr, nullptr};
  uint32_t* d_out[NB] = {nullptr, nullptr, nullptr};
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[NB], ev_k[NB];
  CopyPool pool_in, pool_out;
  bool ok = true;                        //  false: pinned or device memory for the three buffer sets was refused -- the call takes the single-copy path
  HostPipe(size_t chunk_ops, unsigned t_in, unsigned t_out) : chunk(chunk_ops) {
    for (int b = 0; b < NB; b++) { ev_in[b] = nullptr; ev_k[b] = nullptr; }
    for (int b = 0; b < NB && ok; b++) {
      ok = ok && hipHostMalloc(reinterpret_cast<void**>(&h_in[b]), chunk * 12) == hipSuccess;
      ok = ok && hipHostMalloc(reinterpret_cast<void**>(&h_out[b]), chunk * 4) == hipSuccess;
      ok = ok && hipMalloc(reinterpret_cast<void**>(&d_in[b]), chunk * 12) == hipSuccess;
      ok = ok && hipMalloc(reinterpret_cast<void**>(&d_out[b]), chunk * 4) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&ev_in[b], hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&ev_k[b], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return; }
    pool_in.start(t_in);
    pool_out.start(t_out);
  }
  ~HostPipe() {
    for (int b = 0; b < NB; b++) {
      if (h_in[b]) (void)hipHostFree(h_in[b]);
      if (h_out[b]) (void)hipHostFree(h_out[b]);
      if (d_in[b]) (void)hipFree(d_in[b]);
      if (d_out[b]) (void)hipFree(d_out[b]);
      if (ev_in[b]) (void)hipEventDestroy(ev_in[b]);
      if (ev_k[b]) (void)hipEventDestroy(ev_k[b]);
    }
    if (s_in) (void)hipStreamDestroy(s_in);
    if (s_out) (void)hipStreamDestroy(s_out);
  }
};

size_t host_chunk_ops() {              // (read per call: tests set it per handle)
  const char* a = getenv("SMATRIX_HOST_CHUNK_LG");
  const unsigned lg = a ? (unsigned)std::min(26ul, std::max(12ul, strtoul(a, nullptr, 10))) : 21u;
  return (size_t)1 << lg;
}

// n ops from up to three host arrays (in[q] == nullptr: not used), results (when `out`) to a host array; `compute` enqueues the
// kernels of one chunk on the matrix's stream: compute(count, d_a0, d_a1, d_a2, d_out).  Caller holds m->mu.
// the matrix's pipeline for the chunk size in force, or false when its buffers cannot be had (the caller then copies the whole arrays)
bool host_pipe_ready(Matrix* m) {
  if (m->host_pipe && static_cast<HostPipe*>(m->host_pipe)->chunk != host_chunk_ops()) {
    delete static_cast<HostPipe*>(m->host_pipe);
    m->host_pipe = nullptr;
  }
  if (!m->host_pipe) {
    const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
    HostPipe* hp = new HostPipe(host_chunk_ops(), std::min(12u, hw / 2), std::min(8u, std::max(1u, hw / 4)));
    if (!hp->ok) { delete hp; return false; }
    m->host_pipe = hp;
  }
  return true;
}
template <typename F>
void host_pipeline(Matrix* m, size_t n, const uint32_t* const in[3], uint32_t* out, F compute) {
  HostPipe& hp = *static_cast<HostPipe*>(m->host_pipe);      // (host_pipe_ready(m) has been asked)
  const size_t C = hp.chunk, nc = (n + C - 1) / C;
  const int dev = m->device;
  std::mutex mu;
  std::condition_variable cv;
  size_t fed = 0, computed = 0, drained = 0;       // chunks whose upload has been enqueued / whose kernels have been enqueued / returned
  double t_feed_copy = 0, t_feed_wait = 0, t_drain_copy = 0, t_drain_dma = 0, t_main_wait = 0, t_main_compute = 0;   // (SMATRIX_TRACE_ROUNDS)
  const double t_call = mono_s();
  std::thread feeder([&] {
    HIP_OK(hipSetDevice(dev));
    for (size_t k = 0; k < nc; k++) {
      const int b = (int)(k % HostPipe::NB);
      const size_t cnt = std::min(C, n - k * C);
      const double tf0 = mono_s();
      if (k >= (size_t)HostPipe::NB) {
        // the buffers of chunk k - NB: its kernels have been enqueued (they wait for nothing else that reads d_in) ... and have run
        { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return computed >= k - HostPipe::NB + 1; }); }
        HIP_OK(hipEventSynchronize(hp.ev_k[b]));
      }
      const double tf1 = mono_s();
      // The third array -- the amounts of an incr / decr / set call -- is very often ONE value (incr by 1: examples/
      // cf_recommender.c:38-46): the copy threads look while they copy x and y, and a chunk whose amounts are all equal is
      // filled on the device instead of crossing the link (12 -> 8 bytes per op uploaded: the link is what bounds these calls).
      std::atomic<uint32_t> v_differs{1};
      uint32_t v0 = 0;
      std::vector<CopyPool::Slice> v;
      for (int q = 0; q < 2; q++)
        if (in[q]) hp.pool_in.add(v, hp.h_in[b] + (size_t)q * C, in[q] + k * C, cnt * 4);
      if (in[2]) {
        const uint32_t* a = in[2] + k * C;
        v0 = a[0];
        if (a[cnt - 1] == v0 && a[cnt / 2] == v0 && a[cnt / 3] == v0) {        // (three samples first: random amounts end here)
          v_differs.store(0);
          hp.pool_in.add_check(v, a, cnt * 4, v0, &v_differs);
        }
      }
      hp.pool_in.run(std::move(v));
      if (in[2] && v_differs.load()) {
        std::vector<CopyPool::Slice> v2;
        hp.pool_in.add(v2, hp.h_in[b] + 2 * C, in[2] + k * C, cnt * 4);
        hp.pool_in.run(std::move(v2));
      }
      t_feed_wait += tf1 - tf0; t_feed_copy += mono_s() - tf1;
      for (int q = 0; q < 2; q++)
        if (in[q]) HIP_OK(hipMemcpyAsync(hp.d_in[b] + (size_t)q * C, hp.h_in[b] + (size_t)q * C, cnt * 4, hipMemcpyHostToDevice, hp.s_in));
      if (in[2]) {
        if (v_differs.load()) HIP_OK(hipMemcpyAsync(hp.d_in[b] + 2 * C, hp.h_in[b] + 2 * C, cnt * 4, hipMemcpyHostToDevice, hp.s_in));
        else HIP_OK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(hp.d_in[b] + 2 * C), (int)v0, cnt, hp.s_in));
      }
      HIP_OK(hipEventRecord(hp.ev_in[b], hp.s_in));
      { std::lock_guard<std::mutex> l(mu); fed = k + 1; }
      cv.notify_all();
    }
  });
  std::thread drain;
  if (out)
    drain = std::thread([&] {
      HIP_OK(hipSetDevice(dev));
      for (size_t k = 0; k < nc; k++) {
        const int b = (int)(k % HostPipe::NB);
        const size_t cnt = std::min(C, n - k * C);
        { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return computed >= k + 1; }); }
        const double td0 = mono_s();
        HIP_OK(hipStreamWaitEvent(hp.s_out, hp.ev_k[b], 0));
        HIP_OK(hipMemcpyAsync(hp.h_out[b], hp.d_out[b], cnt * 4, hipMemcpyDeviceToHost, hp.s_out));
        HIP_OK(hipStreamSynchronize(hp.s_out));
        const double td1 = mono_s();
        { std::lock_guard<std::mutex> l(mu); drained = k + 1; }       // (d_out of this set may be written again)
        cv.notify_all();
        std::vector<CopyPool::Slice> v;
        hp.pool_out.add(v, out + k * C, hp.h_out[b], cnt * 4);
        hp.pool_out.run(std::move(v));
        t_drain_dma += td1 - td0; t_drain_copy += mono_s() - td1;
      }
    });
  for (size_t k = 0; k < nc; k++) {
    const int b = (int)(k % HostPipe::NB);
    const size_t cnt = std::min(C, n - k * C);
    const double tm0 = mono_s();
    {
      std::unique_lock<std::mutex> l(mu);
      cv.wait(l, [&] { return fed >= k + 1 && (!out || k < (size_t)HostPipe::NB || drained >= k - HostPipe::NB + 1); });
    }
    const double tm1 = mono_s();
    HIP_OK(hipStreamWaitEvent(m->stream, hp.ev_in[b], 0));
    compute(cnt, hp.d_in[b], hp.d_in[b] + C, hp.d_in[b] + 2 * C, hp.d_out[b]);
    HIP_OK(hipEventRecord(hp.ev_k[b], m->stream));
    t_main_wait += tm1 - tm0; t_main_compute += mono_s() - tm1;
    { std::lock_guard<std::mutex> l(mu); computed = k + 1; }
    cv.notify_all();
  }
  feeder.join();
  if (drain.joinable()) drain.join();
  HIP_OK(hipStreamSynchronize(m->stream));
  if (m->trace_rounds)
    fprintf(stderr, "[smatrix] host pipeline: %zu ops in %zu chunks, %.2f ms | feeder: copies %.2f, waits %.2f | caller: waits %.2f, kernels %.2f | drain: device-to-host + waits %.2f, copies %.2f\n",
            n, nc, (mono_s() - t_call) * 1e3, t_feed_copy * 1e3, t_feed_wait * 1e3, t_main_wait * 1e3, t_main_compute * 1e3, t_drain_dma * 1e3, t_drain_copy * 1e3);
}

// ---- C ABI -----------------------------------------------------------------------
extern "C" {

// include/smatrix_batch.h: everything written so far reaches the backing file now (no-op in memory mode)
int smatrix_flush(smatrix_t* self) {
  Matrix* m = M(self);
  if (m->fname.empty() || !self->fd) return 0;
  set_device(m);
  // (like the background flusher: callers of other threads are held up only while the rows are snapshot on the device, not
  //  while they are written; the call itself returns when everything that was dirty at its start is in the file)
  // A BARRIER also against a flush that is in flight: the background flusher takes `dirty`, snapshots the rows,
  // drops the matrix lock and writes under file_mu alone -- a call that arrives during that write finds nothing dirty, yet the
  // rows are not in the file and their CMAP entries not published.  Every turn queues for file_mu FIRST and without the matrix
  // lock (round 6: no caller of the handle is held up behind this call while somebody else's write goes on); once it has the
  // file no flush is in flight, so "nothing dirty" then means everything is in the file.
  for (;;) {
    std::unique_lock<std::mutex> fg(m->file_mu);
    std::unique_lock<std::mutex> g(m->mu);
    if (!m->dirty.exchange(false)) break;
    cache_sync(m, false);
    if (!file_flush(self, m, false, &g)) break;    // everything dirty when this turn began has been written by it
    m->dirty = true;                               // (the snapshot budget was reached: another turn)
  }
  return 0;
}

// include/smatrix_batch.h: the backing file rewritten without leaked blocks (no-op in memory mode)
int smatrix_compact(smatrix_t* self) {
  Matrix* m = M(self);
  // EXPERIMENTAL and outside the drop-in surface (the reference has no counterpart: its files only grow, src/smatrix.c:430-436):
  // the prototype sits behind SMATRIX_EXPERIMENTAL in the header, and the call does nothing unless the process asks for it
  const char* ex = getenv("SMATRIX_EXPERIMENTAL");
  if (!ex || *ex != '1') {
    fprintf(stderr, "libsmatrix: smatrix_compact is experimental; set SMATRIX_EXPERIMENTAL=1 to use it (nothing done)\n");
    return -1;
  }
  if (m->fname.empty() || !self->fd) return 0;
  set_device(m);
  std::lock_guard<std::mutex> fg(m->file_mu);
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, false);
  file_compact(self, m);
  return 0;
}

int smatrix_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n > 0;
}

// Every SMATRIX_* variable the library reads (INTEGRATION.md "Switches" has one line and one test for each).  A variable that
// is set but not in this list -- a typo, or a switch of an earlier round that is gone -- gets ONE warning per process.
static const char* const KNOWN_SWITCHES[] = {
    "SMATRIX_BULK", "SMATRIX_BULK_MIN", "SMATRIX_BULK_PRESIZE_MIN", "SMATRIX_BULK_SHARE", "SMATRIX_CHUNK_POOL_GB", "SMATRIX_CLUSTERED", "SMATRIX_COLD_MIN",
    "SMATRIX_COLD_SHARE", "SMATRIX_COMPACT_AT_CLOSE", "SMATRIX_DEVICE", "SMATRIX_EXPERIMENTAL", "SMATRIX_FAR_JOIN", "SMATRIX_FAR_PLACE", "SMATRIX_FLUSH_EVERY",
    "SMATRIX_FLUSH_MS", "SMATRIX_FLUSH_SNAPSHOT_MB", "SMATRIX_FLUSH_SNAPSHOT_REFUSE", "SMATRIX_FSYNC", "SMATRIX_HINT_LG", "SMATRIX_HIP_LIB", "SMATRIX_HOST_CHUNK_LG",
    "SMATRIX_IO_THREADS", "SMATRIX_IO_WINDOW_MB", "SMATRIX_NO_VMM", "SMATRIX_PEND", "SMATRIX_RCCL_LIB", "SMATRIX_REST_LDS", "SMATRIX_REST_SLICE", "SMATRIX_SCALAR_CACHE",
    "SMATRIX_SCALAR_CACHE_CAP", "SMATRIX_SCRATCH_POOL", "SMATRIX_SET_LOCATE", "SMATRIX_SHARD_FORCE_RCCL", "SMATRIX_SHARD_PLACE", "SMATRIX_SHARD_SHM_MB",
    "SMATRIX_SHARD_TRANSPORT", "SMATRIX_SPEC", "SMATRIX_SPEC_TINY", "SMATRIX_TRACE_ROUNDS",
    "SMATRIX_SHARD_HOST_STAGED",        // (read by libsmatrix_amd/sharded.py, the torch.distributed router: test rigs with two ranks on one GPU)
#ifdef SMX_MEASURE
    "SMATRIX_DBG_AFTER", "SMATRIX_REST_DBG", "SMATRIX_REST_DBG_FROM",
#endif
};
extern "C" char** environ;
static void warn_unknown_switches() {
  static std::once_flag once;
  std::call_once(once, [] {
    for (char** e = environ; e && *e; e++) {
      if (strncmp(*e, "SMATRIX_", 8) != 0) continue;
      const char* eq = strchr(*e, '=');
      const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
      bool known = false;
      for (const char* k : KNOWN_SWITCHES) known |= strlen(k) == len && strncmp(k, *e, len) == 0;
      if (!known) fprintf(stderr, "libsmatrix: the environment sets %.*s, which this library does not read (INTEGRATION.md lists its switches)\n", (int)len, *e);
    }
  });
}

// src/smatrix.c:74-111
smatrix_t* smatrix_open(const char* fname) {
  warn_unknown_switches();
  if (!smatrix_device_available()) {
    fprintf(stderr, "libsmatrix: no HIP device available (this build has no CPU fallback)\n");
    return nullptr;
  }
  smatrix_t* self = static_cast<smatrix_t*>(calloc(1, sizeof(smatrix_t)));
  if (!self) return nullptr;
  Matrix* m = new Matrix();
  self->impl = m;

  int dev = 0;
  const char* env = getenv("SMATRIX_DEVICE");
  if (env && *env) dev = atoi(env);
  else HIP_OK(hipGetDevice(&dev));
  m->device = dev;
  set_device(m);
  // a BLOCKING stream: ordered against the legacy default stream, so host-pointer calls (this stream)
  // and device-pointer calls with hip_stream == NULL (the default stream) never overlap each other
  HIP_OK(hipStreamCreate(&m->stream));
  dev_malloc(&m->d_ctl, sizeof(Ctl));
  HIP_OK(hipHostMalloc(&m->h_ctl, sizeof(Ctl)));
  dev_malloc(&m->d_small, 64);
  HIP_OK(hipHostMalloc(&m->h_small, 64));
  HIP_OK(hipHostMalloc(&m->h_row, 32 + (size_t)SCALAR_ROW_PAIRS_ALLOC * 8));
  if (const char* a = getenv("SMATRIX_SET_LOCATE")) m->set_always_locate = *a == '1';
  if (const char* a = getenv("SMATRIX_BULK_PRESIZE_MIN")) m->fix_presize_min = (uint32_t)strtoul(a, nullptr, 10);
  if (const char* a = getenv("SMATRIX_FLUSH_SNAPSHOT_MB")) m->flush_snapshot = std::max<uint64_t>(1, strtoull(a, nullptr, 10)) << 20;
  if (const char* a = getenv("SMATRIX_SCALAR_CACHE")) m->cache.enabled = *a != '0';
  if (const char* a = getenv("SMATRIX_SCALAR_CACHE_CAP")) m->cache.shard_cap = std::max<size_t>(4, std::min<size_t>(strtoull(a, nullptr, 10), CellCache::SLOTS / 2));   // (tests: constant recycling)
  HIP_OK(hipEventCreate(&m->ev0));
  HIP_OK(hipEventCreate(&m->ev1));
  if (m->grow_fork) {
    HIP_OK(hipStreamCreateWithFlags(&m->helper, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
  }
  m->dir_size = 65536;                               // SMATRIX_CMAP_INITIAL_SIZE, src/smatrix.h:24
  dev_malloc(&m->d_dir, (size_t)m->dir_size * sizeof(DirSlot));
  zero_async(m->d_dir, (size_t)m->dir_size * sizeof(DirSlot), m->stream);
  m->arena.init(dev, 4u << 20, m->stream);
  // the growth passes' buffers start at sizes a steady-state batch of 2^24 ops needs (30 MB in all), so that the first
  // batches that use them do not stop to allocate (a hipFree inside a timed step is a device-wide sync)
  m->map_old.need((size_t)1 << 20);
  m->map_new.need((size_t)1 << 21);
  for (uint32_t c = 0; c < N_CLASSES; c++) ensure_free_cap(m, c, c < 12 ? 1u << 18 : 1u << 12, m->stream);
  {
    // the library's stream-ordered pool (scratch_pool; DevBuf::need_on: scratch of one batch): its first use -- 8 ms to set a
    // pool up -- happens here
    if (hipMemPool_t pool = scratch_pool()) {
      void* warm = nullptr;
      if (hipMallocFromPoolAsync(&warm, 4096, pool, m->stream) == hipSuccess) (void)hipFreeAsync(warm, m->stream);
    }
    (void)hipGetLastError();
  }
  HIP_OK(hipStreamSynchronize(m->stream));
  ctl_push_persistent(m, m->stream);
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grow_lds<1024, GROW_LG2, false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)grow_lds_bytes(GROW_LG2)));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grow_lds<1024, GROW_LG2, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)grow_lds_bytes(GROW_LG2)));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grow_rest_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rest_lds_bytes()));
  if (const char* a = getenv("SMATRIX_REST_LDS")) m->rest_lds = *a != '0';
  if (const char* a = getenv("SMATRIX_FAR_PLACE")) m->far_place = *a != '0';
  if (const char* a = getenv("SMATRIX_PEND")) m->pend_on = *a != '0';
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_far_place<512, FAR_PLACE_SMALL_LG + 1, REST_LDS_MAX_LG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)far_place_lds_bytes(REST_LDS_MAX_LG)));
  if (const char* a = getenv("SMATRIX_REST_SLICE")) m->rest_slice_cells = std::max<uint32_t>(64, (uint32_t)strtoul(a, nullptr, 10));
  if (const char* a = getenv("SMATRIX_FAR_JOIN")) m->far_join = *a != '0';
#ifdef SMX_MEASURE    /* measurement builds only (make HIPCC="hipcc -DSMX_MEASURE"): the event counters of the dense-id kernels, printed at close */
  if (const char* a = getenv("SMATRIX_REST_DBG_FROM")) m->rest_dbg_from = strtoull(a, nullptr, 10);     // (counters from this batch on)
  if (const char* a = getenv("SMATRIX_REST_DBG")) {
    m->rest_dbg_mode = (uint32_t)strtoul(a, nullptr, 10);
    dev_malloc(&m->rest_dbg, 1024);
    HIP_OK(hipMemset(m->rest_dbg, 0, 1024));
    HIP_OK(hipMemcpy(m->arena.base + offsetof(ArenaHead, dbg), &m->rest_dbg, 8, hipMemcpyHostToDevice));
  }
  if (const char* a = getenv("SMATRIX_DBG_AFTER")) m->dbg_after = strtoull(a, nullptr, 10);
#endif
  m->io_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* a = getenv("SMATRIX_BULK")) m->bulk_enabled = *a != '0';
  if (const char* a = getenv("SMATRIX_SPEC")) m->spec_enabled = *a != '0';
  if (const char* a = getenv("SMATRIX_COLD_MIN")) m->cold_min = (uint32_t)strtoul(a, nullptr, 10);
  if (const char* a = getenv("SMATRIX_COLD_SHARE")) m->cold_share = std::max(1u, (uint32_t)strtoul(a, nullptr, 10));
  if (const char* a = getenv("SMATRIX_CLUSTERED")) { m->clustered = *a != '0'; m->clustered_forced = true; }
  if (const char* a = getenv("SMATRIX_HINT_LG")) m->hint_lg = std::min(28u, (uint32_t)strtoul(a, nullptr, 10));
  clustered_sync(m, m->stream);
  if (const char* a = getenv("SMATRIX_SPEC_TINY")) m->spec_tiny = *a == '1';
  if (const char* a = getenv("SMATRIX_BULK_MIN")) m->fix_min = (uint32_t)strtoul(a, nullptr, 10);
  if (const char* a = getenv("SMATRIX_BULK_SHARE")) m->fix_share = std::max(1u, (uint32_t)strtoul(a, nullptr, 10));
  if (const char* a = getenv("SMATRIX_FSYNC")) m->file_fsync = *a == '1';
  if (const char* a = getenv("SMATRIX_COMPACT_AT_CLOSE")) m->compact_at_close = *a == '1' && getenv("SMATRIX_EXPERIMENTAL") && *getenv("SMATRIX_EXPERIMENTAL") == '1';
  if (const char* a = getenv("SMATRIX_FLUSH_EVERY")) m->flush_every = strtoull(a, nullptr, 10);
  if (const char* a = getenv("SMATRIX_FLUSH_MS")) m->flush_ms = strtoull(a, nullptr, 10);
  if (const char* a = getenv("SMATRIX_IO_WINDOW_MB")) m->io_window = std::max<uint64_t>(1, strtoull(a, nullptr, 10)) << 20;
  if (const char* a = getenv("SMATRIX_IO_THREADS")) m->io_threads = std::max(1u, (unsigned)strtoul(a, nullptr, 10));
  if (const char* t = getenv("SMATRIX_TRACE_ROUNDS")) { m->trace_rounds = *t == '1' || *t == '2' || *t == '3'; m->trace_sync = *t == '2'; }
  m->profile = false;                   // (smatrix_profile() switches the kernel timers on)

  if (fname) {
    m->fname = fname;
    if (!file_open_or_create(self, m)) {
      smatrix_close(self);
      return nullptr;
    }
    if (m->clustered) { m->home_on = false; clustered_sync(m, m->stream); }     // (the loaded rows' at-home bitmaps)
    if (m->flush_ms) m->flusher = std::thread(flusher_main, self, m);
  }
  refresh_public(self);
  return self;
}

// src/smatrix.c:113-133: in file mode close is the flush barrier
void smatrix_close(smatrix_t* self) {
  if (!self) return;
  Matrix* m = M(self);
  if (m) {
    set_device(m);
    if (m->rest_dbg) {
      unsigned long long c[128];
      HIP_OK(hipMemcpy(c, m->rest_dbg, 1024, hipMemcpyDeviceToHost));
      fprintf(stderr, "[smatrix] far join, long probes of the wave-per-op pass: not in the table %llu, cell known %llu, absent at the scan %llu\n", c[16], c[17], c[18]);
      if (c[24]) fprintf(stderr, "[smatrix] wave-per-op pass with the join, cycles per op: directory + lane probe %.0f, join look-up %.0f, cooperative probes %.0f, apply / insert %.0f  (%llu ops)\n",
                         c[20] / (double)c[24], c[21] / (double)c[24], c[22] / (double)c[24], c[23] / (double)c[24], 64 * c[24]);
      if (c[28]) fprintf(stderr, "[smatrix] wave-per-op pass with the join: %llu trips of more than 10^5 cycles (%.0f on average, the longest %llu, longest cooperative probe %llu); by row size 2^(4k..): %llu %llu %llu %llu %llu %llu; finished %llu, deferred %llu\n",
                         c[28], c[30] / (double)c[28], c[29], c[31], c[32], c[33], c[34], c[35], c[36], c[37], c[38], c[39]);
      if (c[28]) fprintf(stderr, "[smatrix]   ... their cycles: directory + lane probe %.0f, join look-up %.0f, cooperative probes %.0f, apply / insert %.0f\n",
                                          c[40] / (double)c[28], c[41] / (double)c[28], c[42] / (double)c[28], c[43] / (double)c[28]);
      if (c[98] || c[113]) fprintf(stderr, "[smatrix] claimed inserts: %llu (placed %llu, none left %llu), ticks per insert: ticket %.0f (max %llu), claim %.0f (max %llu); words %.1f (max %llu), tries %.2f | above 10^5 ticks: %llu, ticket %.0f, claim %.0f, words %.1f, tries %.1f | no ticket: %llu, %.0f ticks (max %llu)\n",
              c[98], c[104], c[105], c[96] / (double)std::max(1ull, c[98]), c[99], c[97] / (double)std::max(1ull, c[98]), c[100], c[101] / (double)std::max(1ull, c[98]), c[103], c[102] / (double)std::max(1ull, c[98]),
              c[108], c[109] / (double)std::max(1ull, c[108]), c[110] / (double)std::max(1ull, c[108]), c[111] / (double)std::max(1ull, c[108]), c[112] / (double)std::max(1ull, c[108]),
              c[113], c[114] / (double)std::max(1ull, c[113]), c[115]);
      if (c[116]) fprintf(stderr, "[smatrix] retry's wave-per-op pass: %llu ops, %.0f ticks on average (max %llu); above 10^5 ticks: %llu (%.0f on average) by row size 2^(4k..): %llu %llu %llu %llu %llu %llu; deferred %llu\n",
                          c[116], c[117] / (double)c[116], c[118], c[119], c[120] / (double)std::max(1ull, c[119]), c[121], c[122], c[123], c[124], c[125], c[126], c[127]);
      fprintf(stderr, "[smatrix] k_grow_rest_lds: steps %llu rounds %llu | most steps of a wave %llu, most rounds %llu | trips %llu, most of a wave %llu | per round: losers %.2f blocked %.2f committed %.2f\n",
              c[0], c[1], c[3], c[6], c[4], c[5], c[8] / (double)std::max(1ull, c[1]), c[9] / (double)std::max(1ull, c[1]), c[10] / (double)std::max(1ull, c[1]));
      fprintf(stderr, "[smatrix] k_grow_rest_lds clock ticks, longest of all launches: row %llu = set-up %llu + waves %llu (placing %llu); a wave on average: %.0f (placing %.0f) over %llu waves\n",
              c[50], c[51], c[52], c[53], c[54] / (double)std::max(1ull, c[56]), c[55] / (double)std::max(1ull, c[56]), c[56]);
      fprintf(stderr, "[smatrix] k_grow_rest_lds phases, k ticks max / mean per wave: bitmap %llu / %.0f, cuts %llu / %.0f, cells in front %llu / %.0f, ranges %llu / %.0f, placing (wave) %llu / %.0f\n",
              c[59] >> 10, c[60] / 1024.0 / std::max(1ull, c[56]), c[61] >> 10, c[62] / 1024.0 / std::max(1ull, c[56]), c[57] >> 10, c[58] / 1024.0 / std::max(1ull, c[56]),
              c[63] >> 10, c[49] / 1024.0 / std::max(1ull, c[56]), c[52] >> 10, c[54] / 1024.0 / std::max(1ull, c[56]));
      for (int i = 0; i < 32; i++) if (c[64 + i]) fprintf(stderr, "[smatrix]   batch %%32 == %d: the row wave 0 of which ran longest: %llu k ticks, set-up %llu k, old table 2^%llu\n", i, c[64 + i] >> 40, (c[64 + i] >> 20) & 0xFFFFF, c[64 + i] & 0xFF);
      (void)hipFree(m->rest_dbg);
    }
    if (m->flusher.joinable()) {                       // (it may be inside a flush: close waits for it, then does the last one)
      { std::lock_guard<std::mutex> l(m->fl_mu); m->fl_stop = true; }
      m->fl_cv.notify_all();
      m->flusher.join();
    }
    {
      std::lock_guard<std::mutex> fg(m->file_mu);
      std::lock_guard<std::mutex> g(m->mu);
      cache_sync(m, true);
      if (!m->fname.empty() && self->fd && m->compact_at_close) file_compact(self, m);
      else if (!m->fname.empty() && self->fd && m->dirty.exchange(false)) file_flush(self, m, false, nullptr);   // a matrix that was only read has nothing to persist
      (void)hipStreamSynchronize(m->stream);
      PhaseClock clk(m->trace_rounds, "close");
      m->arena.destroy();
      clk.lap("arena unmapped");
      if (m->d_dir) (void)hipFree(m->d_dir);
      if (m->d_hints) (void)hipFree(m->d_hints);
      if (m->d_ctl) (void)hipFree(m->d_ctl);
      if (m->h_ctl) (void)hipHostFree(m->h_ctl);
      if (m->d_small) (void)hipFree(m->d_small);
      if (m->h_small) (void)hipHostFree(m->h_small);
      if (m->h_row) (void)hipHostFree(m->h_row);
      m->row_ret.release();
      delete static_cast<HostPipe*>(m->host_pipe);
      m->far_tab.release(); m->far_unit_row.release(); m->far_zeros.release(); m->far_occ.release(); m->far_occ0.release(); m->far_clm.release(); m->far_rcnt.release(); m->far_bucket.release(); m->far_prows.release(); m->far_bloom.release(); m->far_unit_info.release(); m->big_list.release(); 
      m->pend_rec.release(); m->pend_keys.release(); m->task_of.release(); m->pend_ctl.release(); m->pend_hash.release();
      for (auto& d : m->defer) d.release();
      for (uint32_t c = 0; c < N_CLASSES; c++)
        if (m->fl.list[c]) (void)hipFree(m->fl.list[c]);
      m->cold_set.release(); m->cold_keys[0].release(); m->cold_keys[1].release(); m->cold_keys[2].release();
      m->cold_idx[0].release(); m->cold_idx[1].release(); m->cold_zero.release();
      m->fx_cnt.release(); m->fx_cur.release(); m->fx_pos.release(); m->fx_touched.release(); m->fx_where.release(); m->fx_grouped.release(); m->fx_rank.release(); m->fx_excl.release(); m->fx_tiles.release();
      graveyard().empty();                         // (the buffers this and other matrices of the process have outgrown)
      m->tasks.release(); m->klist.release(); m->rebal.release(); m->map_old.release(); m->map_new.release(); m->disp_mask.release(); m->rest_tab.release(); m->cellp.release();
      m->sx.release(); m->sy.release(); m->sv.release(); m->so.release(); m->soff.release(); m->big.release(); m->seg.release(); m->ent_idx.release();
      get_timing_resolve(m, 0);
      for (hipEvent_t e : m->ev_free) (void)hipEventDestroy(e);
      if (m->ev0) (void)hipEventDestroy(m->ev0);
      if (m->ev1) (void)hipEventDestroy(m->ev1);
      if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
      if (m->ev_join) (void)hipEventDestroy(m->ev_join);
      if (m->helper) (void)hipStreamDestroy(m->helper);
      if (m->flush_stream) (void)hipStreamDestroy(m->flush_stream);
      if (m->stream) (void)hipStreamDestroy(m->stream);
      clk.lap("buffers freed");
    }
    if (self->fd) {
      PhaseClock clk(m->trace_rounds, "close");
      close(self->fd);
      clk.lap("close(fd)");
    }
    file_index_free(m);
    delete m;
  }
  free(self);
}

int smatrix_apply_batch_dev(smatrix_t* self, int op, size_t n, const uint32_t* d_x,
                            const uint32_t* d_y, const uint32_t* d_v, uint32_t* d_out,
                            void* hip_stream) {
  Matrix* m = M(self);
  set_device(m);
  CkptAfter ckpt(self);                                  // (a checkpoint that falls due is taken after m->mu is released)
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, op != OP_GET);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
  // results not wanted: the kernels that always write them get the library's own staging buffer
  if (!d_out && n) { m->so.need(n); d_out = m->so.p; m->no_ret = true; }
  apply_dev_locked(self, op, n, d_x, d_y, d_v, d_out, s);
  m->no_ret = false;
  if (!hip_stream) HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_apply_packed_dev(smatrix_t* self, int op, size_t n, const uint32_t* d_records, uint32_t width,
                             uint32_t* d_out, void* hip_stream) {
  if (width != 2 && width != 3) return -1;
  if (op != OP_GET && width != 3) return -1;               // writes need a value
  Matrix* m = M(self);
  set_device(m);
  CkptAfter ckpt(self);                                  // (a checkpoint that falls due is taken after m->mu is released)
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, op != OP_GET);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
  m->in_stride = width;
  if (!d_out && n) { m->so.need(n); d_out = m->so.p; m->no_ret = true; }
  apply_dev_locked(self, op, n, d_records, d_records + 1, d_records + 2, d_out, s);
  m->no_ret = false;
  m->in_stride = 1;
  if (!hip_stream) HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_apply_batch(smatrix_t* self, int op, size_t n, const uint32_t* x, const uint32_t* y,
                        const uint32_t* v, uint32_t* out) {
  if (n == 0) return 0;
  if (n >= (1ull << 32)) smx_die("batch too large (n must be < 2^32)");     // before anything is staged or copied
  Matrix* m = M(self);
  set_device(m);
  CkptAfter ckpt(self);                                  // (a checkpoint that falls due is taken after m->mu is released)
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, op != OP_GET);
  hipStream_t s = m->stream;
  if (n >= 2 * host_chunk_ops() && (op == OP_GET || v) && host_pipe_ready(m)) {
    // a large call: chunks through pinned memory, upload / kernels / return overlapped (host_pipeline)
    const uint32_t* in[3] = {x, y, op != OP_GET ? v : nullptr};
    m->no_ret = out == nullptr;
    const uint64_t batches0 = m->st.batches;
    host_pipeline(m, n, in, out, [&](size_t cnt, uint32_t* dx, uint32_t* dy, uint32_t* dv, uint32_t* dout) {
      apply_dev_locked(self, op, cnt, dx, dy, dv, dout, s);
    });
    m->no_ret = false;
    if (op != OP_GET) {
      //  the call is ONE batch to the caller's statistics and to SMATRIX_FLUSH_EVERY, however many chunks it ran in
      m->st.batches = batches0 + 1;
      m->ckpt_due = m->flush_every && self->fd && m->st.batches % m->flush_every == 0;
    }
    return 0;
  }
  m->sx.need(n); m->sy.need(n); m->so.need(n);
  HIP_OK(hipMemcpyAsync(m->sx.p, x, n * 4, hipMemcpyHostToDevice, s));
  HIP_OK(hipMemcpyAsync(m->sy.p, y, n * 4, hipMemcpyHostToDevice, s));
  if (op != OP_GET) {
    m->sv.need(n);
    HIP_OK(hipMemcpyAsync(m->sv.p, v, n * 4, hipMemcpyHostToDevice, s));
  }
  m->no_ret = out == nullptr;
  apply_dev_locked(self, op, n, m->sx.p, m->sy.p, m->sv.p, m->so.p, s);
  m->no_ret = false;
  if (out) HIP_OK(hipMemcpyAsync(out, m->so.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_get_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y, uint32_t* out) {
  return smatrix_apply_batch(self, OP_GET, n, x, y, nullptr, out);
}
int smatrix_set_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y,
                      const uint32_t* v, uint32_t* out) {
  return smatrix_apply_batch(self, OP_SET, n, x, y, v, out);
}
int smatrix_incr_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y,
                       const uint32_t* v, uint32_t* out) {
  return smatrix_apply_batch(self, OP_INCR, n, x, y, v, out);
}
int smatrix_decr_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint32_t* y,
                       const uint32_t* v, uint32_t* out) {
  return smatrix_apply_batch(self, OP_DECR, n, x, y, v, out);
}

int smatrix_rowlen_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_x, uint32_t* d_out,
                             void* hip_stream) {
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
  hipLaunchKernelGGL(k_rowlen, dim3(blocks_for(n)), dim3(256), 0, s, m->d_dir, m->dir_size - 1,
                     m->arena.base, (uint32_t)n, d_x, d_out);
  HIP_OK(hipGetLastError());
  if (!hip_stream) HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_rowlen_batch(smatrix_t* self, size_t n, const uint32_t* x, uint32_t* out) {
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  hipStream_t s = m->stream;
  if (n >= 2 * host_chunk_ops() && host_pipe_ready(m)) {
    const uint32_t* in[3] = {x, nullptr, nullptr};
    host_pipeline(m, n, in, out, [&](size_t cnt, uint32_t* dx, uint32_t*, uint32_t*, uint32_t* dout) {
      hipLaunchKernelGGL(k_rowlen, dim3(blocks_for(cnt)), dim3(256), 0, s, m->d_dir, m->dir_size - 1, m->arena.base, (uint32_t)cnt, dx, dout);
      HIP_OK(hipGetLastError());
    });
    return 0;
  }
  m->sx.need(n); m->so.need(n);
  HIP_OK(hipMemcpyAsync(m->sx.p, x, n * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_rowlen, dim3(blocks_for(n)), dim3(256), 0, s, m->d_dir, m->dir_size - 1,
                     m->arena.base, (uint32_t)n, m->sx.p, m->so.p);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(out, m->so.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}

namespace {
// the noted (large) rows of a getrow launch: plan their segments, count the cut rows' segments, write
void launch_getrow_big(Matrix* m, hipStream_t s, uint32_t n, uint32_t grid, const uint32_t* x, const uint64_t* off,
                       uint64_t* ret, uint32_t* counts, const uint32_t* big) {
  // seg: [0 .. n] first segment of each noted row, then one count per segment.  Every noted occurrence of a row has
  // one segment; `budget` more are there for cutting giant rows (all DISTINCT rows together have at most
  // mapped / 8 / GETROW_SEG of them; a batch that names one giant row over and over runs out of budget and walks the
  // later occurrences whole, k_getrow_plan -- ADVICE r2: the array used to be overrun in that case)
  const uint32_t budget = (uint32_t)std::min<uint64_t>(m->arena.mapped / 8 / GETROW_SEG + 2, 1u << 24);
  const size_t segs = (size_t)n + budget;
  m->seg.need((size_t)n + 1 + segs);
  uint32_t* seg_start = m->seg.p;
  uint32_t* seg_cnt = m->seg.p + n + 1;
  hipLaunchKernelGGL(k_getrow_plan, dim3(1), dim3(1024), 0, s, m->d_dir, m->dir_size - 1, x, big, seg_start, budget);
  hipLaunchKernelGGL(k_getrow_big<true>, dim3(grid), dim3(1024), 0, s, m->d_dir, m->dir_size - 1, m->arena.base,
                     x, off, ret, counts, big, seg_start, seg_cnt);
  hipLaunchKernelGGL(k_getrow_big<false>, dim3(grid), dim3(1024), 0, s, m->d_dir, m->dir_size - 1, m->arena.base,
                     x, off, ret, counts, big, seg_start, seg_cnt);
  HIP_OK(hipGetLastError());
}

void launch_getrow(Matrix* m, hipStream_t s, uint32_t n, const uint32_t* x, const uint64_t* off,
                   uint64_t* ret, uint32_t* counts) {
  m->big.need((size_t)n + 1);
  HIP_OK(hipMemsetAsync(m->big.p, 0, 4, s));
  uint32_t grid = std::min<uint32_t>(blocks_for((uint64_t)n * 64), 16384);
  constexpr int variant = 0;          // (the variants of profiles/r03_getrow_variants.txt were measured with a switch here; 0 is what ships)
#define GR(A, X, D) hipLaunchKernelGGL((k_getrow<A, X, D>), dim3(grid), dim3(256), 0, s, m->d_dir, m->dir_size - 1, m->arena.base, n, x, off, ret, counts, m->big.p)
  switch (variant) {
    case 1: GR(1, false, 0); break;      // round 2's shape: one step ahead, workgroups as numbered
    case 2: GR(1, true, 0); break;
    case 3: GR(4, true, 0); break;
    case 4: GR(2, false, 0); break;
    case 5: GR(2, true, 1); break;       // no pair stores
    case 6: GR(2, true, 2); break;       // no cell loads
    default: GR(2, true, 0); break;
  }
#undef GR
  // rows of more than 8192 cells were only noted down: a workgroup per row -- per segment of a giant row -- (with
  // none noted the three launches are a few idle workgroups)
  const uint64_t most = (uint64_t)n + m->arena.mapped / 8 / GETROW_SEG;
  launch_getrow_big(m, s, n, (uint32_t)std::min<uint64_t>(most, 2048), x, off, ret, counts, m->big.p);
}
}  // namespace

int smatrix_getrow_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_x,
                             const uint64_t* d_offsets, uint32_t* d_ret, uint32_t* d_counts,
                             void* hip_stream) {
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, false);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
  launch_getrow(m, s, (uint32_t)n, d_x, d_offsets, reinterpret_cast<uint64_t*>(d_ret), d_counts);
  if (!hip_stream) HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_getrow_batch(smatrix_t* self, size_t n, const uint32_t* x, const uint64_t* offsets,
                         uint32_t* ret, uint32_t* counts) {
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, false);
  hipStream_t s = m->stream;
  const uint64_t total = offsets[n];
  m->sx.need(n); m->so.need(n); m->soff.need(n + 1);
  DevBuf<uint64_t>& dret = m->row_ret;                       // pooled: no hipMalloc/hipFree per call
  dret.need(std::max<uint64_t>(total, 1));
  HIP_OK(hipMemcpyAsync(m->sx.p, x, n * 4, hipMemcpyHostToDevice, s));
  HIP_OK(hipMemcpyAsync(m->soff.p, offsets, (n + 1) * 8, hipMemcpyHostToDevice, s));
  launch_getrow(m, s, (uint32_t)n, m->sx.p, m->soff.p, dret.p, m->so.p);
  HIP_OK(hipMemcpyAsync(counts, m->so.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  // copy only what was written per row? rows are packed by the caller's offsets: one copy
  if (total) HIP_OK(hipMemcpy(ret, dret.p, total * 8, hipMemcpyDeviceToHost));
  if (dret.cap > ((size_t)64 << 20)) dret.release();         // (a rare giant request does not pin HBM for good)
  return 0;
}

// ---- CF-recommender read path (include/smatrix_batch.h) ------------------------------------------
int smatrix_cf_neighbors_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_items,
                                   const uint64_t* d_offsets, uint32_t* d_ids, double* d_scores,
                                   uint32_t* d_counts, void* hip_stream) {
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, false);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
  uint32_t grid = std::min<uint32_t>(blocks_for((uint64_t)n * 64), 16384);
  hipLaunchKernelGGL(k_cf_neighbors, dim3(grid), dim3(256), 0, s, m->d_dir, m->dir_size - 1, m->arena.base,
                     (uint32_t)n, d_items, d_offsets, d_ids, d_scores, d_counts);
  HIP_OK(hipGetLastError());
  if (!hip_stream) HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_cf_neighbors_batch(smatrix_t* self, size_t n, const uint32_t* items, const uint64_t* offsets,
                               uint32_t* ids, double* scores, uint32_t* counts) {
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  const uint64_t total = offsets[n];
  uint32_t *d_items = nullptr, *d_ids = nullptr, *d_counts = nullptr;
  uint64_t* d_off = nullptr;
  double* d_scores = nullptr;
  dev_malloc(&d_items, n * 4);
  dev_malloc(&d_off, (n + 1) * 8);
  dev_malloc(&d_counts, n * 4);
  dev_malloc(&d_ids, std::max<uint64_t>(total, 1) * 4);
  dev_malloc(&d_scores, std::max<uint64_t>(total, 1) * 8);
  HIP_OK(hipMemset(d_ids, 0, std::max<uint64_t>(total, 1) * 4));        // slots beyond a row's count read 0
  HIP_OK(hipMemset(d_scores, 0, std::max<uint64_t>(total, 1) * 8));
  HIP_OK(hipMemcpy(d_items, items, n * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_off, offsets, (n + 1) * 8, hipMemcpyHostToDevice));
  smatrix_cf_neighbors_batch_dev(self, n, d_items, d_off, d_ids, d_scores, d_counts, nullptr);
  HIP_OK(hipMemcpy(counts, d_counts, n * 4, hipMemcpyDeviceToHost));
  if (total) {
    HIP_OK(hipMemcpy(ids, d_ids, total * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(scores, d_scores, total * 8, hipMemcpyDeviceToHost));
  }
  (void)hipFree(d_items); (void)hipFree(d_off); (void)hipFree(d_counts); (void)hipFree(d_ids); (void)hipFree(d_scores);
  return 0;
}

static void row_info_locked(Matrix* m, uint32_t x, uint32_t* four);

// ---- the reference's scalar entry points ----------------------------------------------------------
// One scalar op on the device, with the matrix lock held.
static uint32_t scalar_one_locked(smatrix_t* self, Matrix* m, int op, uint32_t x, uint32_t y, uint32_t v) {
  hipStream_t s = m->stream;
  if (op != OP_GET && y == 0) cache_sync(m, true);       // a y = 0 write can cut probe chains: nothing stays mirrored
  // fast path: one launch, result written by the kernel into pinned host memory, one sync.
  // (set keeps its value write inside apply_row; a one-op batch has no duplicates to resolve)
  {
    volatile uint32_t* res = m->h_small + 8;
    const dim3 one(1);
    switch (op) {
      case OP_GET:  hipLaunchKernelGGL((k_scalar<OP_GET>), one, one, 0, s, m->d_dir, m->dir_size - 1, m->arena.base, x, y, v, res); break;
      case OP_SET:  hipLaunchKernelGGL((k_scalar<OP_SET>), one, one, 0, s, m->d_dir, m->dir_size - 1, m->arena.base, x, y, v, res); break;
      case OP_INCR: hipLaunchKernelGGL((k_scalar<OP_INCR>), one, one, 0, s, m->d_dir, m->dir_size - 1, m->arena.base, x, y, v, res); break;
      default:      hipLaunchKernelGGL((k_scalar<OP_DECR>), one, one, 0, s, m->d_dir, m->dir_size - 1, m->arena.base, x, y, v, res); break;
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(s));
    if (!res[1]) {
      if (op != OP_GET) { m->st.batches++; m->dirty = true; }
      // after a write the cell exists and holds res[0]; a get proves existence only through a non-zero value
      if (op != OP_GET || res[0] != 0) m->cache.put(x, y, res[0]);
      return res[0];
    }
  }
  m->h_small[0] = x; m->h_small[1] = y; m->h_small[2] = v;
  HIP_OK(hipMemcpyAsync(m->d_small, m->h_small, 12, hipMemcpyHostToDevice, s));
  apply_dev_locked(self, op, 1, m->d_small, m->d_small + 1, m->d_small + 2, m->d_small + 3, s);
  HIP_OK(hipMemcpyAsync(m->h_small + 3, m->d_small + 3, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  if (op != OP_GET) m->cache.put(x, y, m->h_small[3]);
  return m->h_small[3];
}

// The reference's scalar calls are thread-safe and callers (JVM threads, src/smatrix_jni.c) issue them
// concurrently.  Calls on mirrored cells never get here (CellCache).  For the others a device round trip per call
// would serialise the threads at ~15 us each, so concurrent callers are COMBINED: every caller queues its op;
// whoever finds no combiner active becomes it, takes everything queued so far and runs it as one batch per op kind
// (any order among concurrent calls is a legal serialisation; each thread's own calls stay ordered because a thread
// has one call in flight).  A lone caller pays nothing extra: a queue of one goes down the one-launch fast path.
static uint32_t scalar_op(smatrix_t* self, int op, uint32_t x, uint32_t y, uint32_t v) {
  Matrix* m = M(self);
  uint32_t hit;
  if (m->cache.apply(op, x, y, v, &hit)) {
    if (op != OP_GET) m->dirty = true;
    return hit;
  }
  ScalarReq req{op, x, y, v, 0, false};
  std::unique_lock<std::mutex> ql(m->qmu);
  m->queue.push_back(&req);
  if (m->combining) {
    m->qcv.wait(ql, [&] { return req.done || !m->combining; });
    if (req.done) return req.result;
  }
  // become the combiner
  m->combining = true;
  std::vector<ScalarReq*> work, dev;
  while (!m->queue.empty()) {
    work.clear();
    work.swap(m->queue);
    ql.unlock();
    {
      set_device(m);
      CkptAfter ckpt(self);                                  // (a checkpoint that falls due is taken after m->mu is released)
      std::lock_guard<std::mutex> g(m->mu);
      // a cell may have been mirrored while this request waited for the lock (another thread's device op on it):
      // look again -- under the matrix lock mirrored-ness cannot change under us
      dev.clear();
      for (ScalarReq* r : work)
        if (!m->cache.apply(r->op, r->x, r->y, r->v, &r->result)) dev.push_back(r);
        else if (r->op != OP_GET) m->dirty = true;
      if (dev.size() == 1) {
        ScalarReq* r = dev[0];
        r->result = scalar_one_locked(self, m, r->op, r->x, r->y, r->v);
      } else if (!dev.empty()) {
        hipStream_t s = m->stream;
        bool y0_write = false;
        for (ScalarReq* r : dev) y0_write |= r->op != OP_GET && r->y == 0;
        cache_sync(m, y0_write);
        for (int kind = 0; kind < 4; kind++) {
          size_t k = 0;
          for (ScalarReq* r : dev) k += r->op == kind;
          if (!k) continue;
          std::vector<uint32_t> hx(k), hy(k), hv(k), ho(k);
          size_t i = 0;
          for (ScalarReq* r : dev)
            if (r->op == kind) { hx[i] = r->x; hy[i] = r->y; hv[i] = r->v; i++; }
          m->sx.need(k); m->sy.need(k); m->sv.need(k); m->so.need(k);
          HIP_OK(hipMemcpyAsync(m->sx.p, hx.data(), k * 4, hipMemcpyHostToDevice, s));
          HIP_OK(hipMemcpyAsync(m->sy.p, hy.data(), k * 4, hipMemcpyHostToDevice, s));
          HIP_OK(hipMemcpyAsync(m->sv.p, hv.data(), k * 4, hipMemcpyHostToDevice, s));
          apply_dev_locked(self, kind, k, m->sx.p, m->sy.p, m->sv.p, m->so.p, s);
          HIP_OK(hipMemcpyAsync(ho.data(), m->so.p, k * 4, hipMemcpyDeviceToHost, s));
          HIP_OK(hipStreamSynchronize(s));
          i = 0;
          for (ScalarReq* r : dev)
            if (r->op == kind) r->result = ho[i++];
        }
        // cells written exactly once in this combined group now exist with a known value: mirror them
        // (a cell named twice has two results of which only "some serialisation" is known -- left to the next call)
        std::vector<uint64_t> keys;
        keys.reserve(dev.size());
        for (ScalarReq* r : dev) keys.push_back((uint64_t)r->x << 32 | r->y);
        std::sort(keys.begin(), keys.end());
        // (not when the group held a y = 0 write: it may have turned a row's (0,v) cell into an empty one and cut the
        //  probe chain of a key an earlier kind of the same group touched -- the mirror would then vouch for a cell
        //  the reference no longer finds)
        for (ScalarReq* r : dev) {
          if (r->op == OP_GET || y0_write) continue;
          const uint64_t key = (uint64_t)r->x << 32 | r->y;
          auto range = std::equal_range(keys.begin(), keys.end(), key);
          if (range.second - range.first == 1) m->cache.put(r->x, r->y, r->result);
        }
      }
    }
    ql.lock();
    for (ScalarReq* r : work) r->done = true;
    m->qcv.notify_all();
  }
  m->combining = false;
  m->qcv.notify_all();
  return req.result;
}

uint32_t smatrix_get(smatrix_t* self, uint32_t x, uint32_t y) { return scalar_op(self, OP_GET, x, y, 0); }
uint32_t smatrix_set(smatrix_t* self, uint32_t x, uint32_t y, uint32_t value) { return scalar_op(self, OP_SET, x, y, value); }
uint32_t smatrix_incr(smatrix_t* self, uint32_t x, uint32_t y, uint32_t value) { return scalar_op(self, OP_INCR, x, y, value); }
uint32_t smatrix_decr(smatrix_t* self, uint32_t x, uint32_t y, uint32_t value) { return scalar_op(self, OP_DECR, x, y, value); }

// src/smatrix.c:212-223.  rmap->used does not depend on mirrored VALUES, so nothing is written back first.
uint32_t smatrix_rowlen(smatrix_t* self, uint32_t x) {
  uint32_t out = 0;
  smatrix_rowlen_batch(self, 1, &x, &out);
  return out;
}

// src/smatrix.c:189-210: ret_len counts BYTES and the loop stops once pairs*8 >= ret_len,
// so a non-empty row always yields at least one pair (S4).
// ONE device round trip for rows of up to SCALAR_ROW_PAIRS pairs (what the bindings ask for, smatrix_jni.c:130-139):
// the row kernel writes its count and the pairs straight into pinned host memory.
static constexpr uint32_t SCALAR_ROW_PAIRS = 8192;      // 64 KB pinned
uint32_t smatrix_getrow(smatrix_t* self, uint32_t x, uint32_t* ret, size_t ret_len) {
  uint64_t cap = (ret_len + 7) / 8;
  if (cap == 0) cap = 1;
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, false);
  hipStream_t s = m->stream;
  // h_row: [0] count, [1] x, [2..3] big list {n, first}, [4..7] offsets {0, cap} as u64, then the pairs
  uint32_t* h = m->h_row;
  uint64_t* offs = reinterpret_cast<uint64_t*>(h + 4);
  uint64_t* pairs = reinterpret_cast<uint64_t*>(h + 8);
  const uint64_t want = std::min<uint64_t>(cap, 0xffffffffull);
  if (want <= SCALAR_ROW_PAIRS) {
    h[0] = 0; h[1] = x; h[2] = 0; h[3] = 0;
    offs[0] = 0; offs[1] = want;
    hipLaunchKernelGGL((k_getrow<2, false, 0>), dim3(1), dim3(64), 0, s, m->d_dir, m->dir_size - 1, m->arena.base, 1u, h + 1,
                       offs, pairs, h, h + 2);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(s));
    if (h[2]) {                                           // a row of more than 8192 cells: the workgroup-per-row kernel
      launch_getrow_big(m, s, 1, 256, h + 1, offs, pairs, h, h + 2);
      HIP_OK(hipStreamSynchronize(s));
    }
    const uint32_t count = h[0];
    memcpy(ret, pairs, (size_t)count * 8);
    return count;
  }
  // a larger buffer: bounded by the row's size, through the pooled device buffer
  uint32_t f[4];
  row_info_locked(m, x, f);
  if (!f[0]) return 0;
  const uint64_t room = std::min<uint64_t>(want, f[1]);
  if (room == 0) return 0;
  h[0] = 0; h[1] = x; h[2] = 0; h[3] = 0;
  offs[0] = 0; offs[1] = room;
  m->row_ret.need(room);
  launch_getrow(m, s, 1, h + 1, offs, m->row_ret.p, h);
  HIP_OK(hipStreamSynchronize(s));
  const uint32_t count = h[0];
  if (count) HIP_OK(hipMemcpy(ret, m->row_ret.p, (size_t)count * 8, hipMemcpyDeviceToHost));
  if (m->row_ret.cap > ((size_t)64 << 20)) m->row_ret.release();
  return count;
}

// the k best neighbours per item (k <= 64); ids / scores hold n * k entries, item i's at [i*k, i*k + counts[i])
int smatrix_cf_topk_batch_dev(smatrix_t* self, size_t n, const uint32_t* d_items, uint32_t k, uint32_t* d_ids,
                              double* d_scores, uint32_t* d_counts, void* hip_stream) {
  if (k == 0 || k > 64) return -1;
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, false);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
  uint32_t grid = std::min<uint32_t>(blocks_for((uint64_t)n * 64), 16384);
  hipLaunchKernelGGL(k_cf_topk, dim3(grid), dim3(256), 0, s, m->d_dir, m->dir_size - 1, m->arena.base, (uint32_t)n, d_items, k,
                     d_ids, d_scores, d_counts);
  HIP_OK(hipGetLastError());
  if (!hip_stream) HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_cf_topk_batch(smatrix_t* self, size_t n, const uint32_t* items, uint32_t k, uint32_t* ids, double* scores,
                          uint32_t* counts) {
  if (k == 0 || k > 64) return -1;
  if (n == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  uint32_t *d_items = nullptr, *d_ids = nullptr, *d_counts = nullptr;
  double* d_scores = nullptr;
  dev_malloc(&d_items, n * 4);
  dev_malloc(&d_counts, n * 4);
  dev_malloc(&d_ids, n * k * 4);
  dev_malloc(&d_scores, n * k * 8);
  HIP_OK(hipMemset(d_ids, 0, n * k * 4));
  HIP_OK(hipMemset(d_scores, 0, n * k * 8));
  HIP_OK(hipMemcpy(d_items, items, n * 4, hipMemcpyHostToDevice));
  smatrix_cf_topk_batch_dev(self, n, d_items, k, d_ids, d_scores, d_counts, nullptr);
  HIP_OK(hipMemcpy(counts, d_counts, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(ids, d_ids, n * k * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(scores, d_scores, n * k * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d_items); (void)hipFree(d_counts); (void)hipFree(d_ids); (void)hipFree(d_scores);
  return 0;
}

// ---- CF-recommender write path (examples/cf_recommender.c:36-47) ------------------------------------------------
// n_sessions sessions, session s = ids[offsets[s] .. offsets[s+1]); d_op_offsets[s] = sum of L*L over the sessions before s
// (n_sessions + 1 entries, the last one = total_ops).  The L*L incr ops of every session are generated on the device in
// chunks of CF_IMPORT_CHUNK ops and applied as ordinary incr batches.
static constexpr uint64_t CF_IMPORT_CHUNK = 1ull << 25;
int smatrix_cf_import_sessions_dev(smatrix_t* self, size_t n_sessions, const uint64_t* d_offsets, const uint32_t* d_ids,
                                   const uint64_t* d_op_offsets, uint64_t total_ops, void* hip_stream) {
  if (n_sessions == 0 || total_ops == 0) return 0;
  if (n_sessions > 0xffffffffull) return -1;
  Matrix* m = M(self);
  set_device(m);
  CkptAfter ckpt(self);                                  // (a checkpoint that falls due is taken after m->mu is released)
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, true);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the legacy default stream
  const size_t chunk = (size_t)std::min<uint64_t>(total_ops, CF_IMPORT_CHUNK);
  m->sx.need(chunk); m->sy.need(chunk); m->sv.need(chunk); m->so.need(chunk);
  for (uint64_t t0 = 0; t0 < total_ops; t0 += CF_IMPORT_CHUNK) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(CF_IMPORT_CHUNK, total_ops - t0);
    hipLaunchKernelGGL(k_cf_expand, dim3(blocks_for(n)), dim3(256), 0, s, t0, n, (uint32_t)n_sessions, d_offsets, d_ids,
                       d_op_offsets, m->sx.p, m->sy.p, m->sv.p);
    HIP_OK(hipGetLastError());
    m->no_ret = true;
    apply_dev_locked(self, OP_INCR, n, m->sx.p, m->sy.p, m->sv.p, m->so.p, s);
    m->no_ret = false;
  }
  if (!hip_stream) HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_cf_import_sessions(smatrix_t* self, size_t n_sessions, const uint64_t* offsets, const uint32_t* ids) {
  if (n_sessions == 0) return 0;
  Matrix* m = M(self);
  set_device(m);
  std::vector<uint64_t> op_off(n_sessions + 1, 0);
  for (size_t i = 0; i < n_sessions; i++) {
    if (offsets[i + 1] < offsets[i]) return -1;
    const uint64_t L = offsets[i + 1] - offsets[i];
    op_off[i + 1] = op_off[i] + L * L;
  }
  const uint64_t n_ids = offsets[n_sessions] - offsets[0], total = op_off[n_sessions];
  if (total == 0) return 0;
  uint64_t *d_off = nullptr, *d_op = nullptr;
  uint32_t* d_ids = nullptr;
  dev_malloc(&d_off, (n_sessions + 1) * 8);
  dev_malloc(&d_op, (n_sessions + 1) * 8);
  dev_malloc(&d_ids, std::max<uint64_t>(n_ids, 1) * 4);
  std::vector<uint64_t> rel(n_sessions + 1);
  for (size_t i = 0; i <= n_sessions; i++) rel[i] = offsets[i] - offsets[0];
  HIP_OK(hipMemcpy(d_off, rel.data(), (n_sessions + 1) * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_op, op_off.data(), (n_sessions + 1) * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_ids, ids + offsets[0], n_ids * 4, hipMemcpyHostToDevice));
  const int rc = smatrix_cf_import_sessions_dev(self, n_sessions, d_off, d_ids, d_op, total, nullptr);
  (void)hipFree(d_off); (void)hipFree(d_op); (void)hipFree(d_ids);
  return rc;
}

// Capacity hint (like vector::reserve): map at least `bytes` of row arena now, so that growth steps -- calls into the
// driver, which can block for seconds when it still has freed memory to wipe (ChunkPool) -- do not fall into a
// latency-sensitive phase later.  Changes nothing observable; 0 on success.
int smatrix_reserve(smatrix_t* self, uint64_t bytes) {
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  if (m->arena.vmm && bytes > m->arena.reserved) bytes = m->arena.reserved;
  if (bytes > m->arena.mapped) m->arena.grow_to(bytes, m->arena_next * UNIT_BYTES, m->stream);
  HIP_OK(hipStreamSynchronize(m->stream));
  return 0;
}

// physical chunks kept from closed matrices (ChunkPool) go back to the driver now
void smatrix_release_cached_memory(void) {
  graveyard().empty();
  chunk_pool().trim();
  int dev = 0;
  hipMemPool_t pool = nullptr;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);
  (void)hipGetLastError();
}

// ---- introspection -------------------------------------------------------------------
void smatrix_stats(smatrix_t* self, smatrix_stats_t* out) {
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  get_timing_resolve(m, 0);
  m->st.rows = m->dir_used;
  m->st.dir_slots = m->dir_size;
  m->st.arena_units = m->arena_next;
  m->st.arena_free_units = 0;
  for (uint32_t c = 0; c < N_CLASSES; c++)
    if (m->free_cnt[c] > 0) m->st.arena_free_units += (uint64_t)m->free_cnt[c] * block_units(c + ROW_FIRST_LG);
  m->st.arena_mapped = m->arena.mapped;
  m->st.scalar_cache_hits = m->cache.hits_total();
  m->st.scalar_cache_flushes = m->cache.flushes.load();
  m->st.scalar_cache_flushed_cells = m->cache.flushed_cells.load();
  m->st.file_leaked_bytes = m->file_index ? file_leaked(m) : 0;
  m->st.file_flushes = m->file_flushes_done.load();
  m->st.file_rows_written = m->file_rows_written_done.load();
  m->st.file_bg_flushes = m->file_bg_flushes_done.load();
  m->st.clustered_mode = m->clustered ? 1 : 0;
  *out = m->st;
}
//  the same for a caller compiled against an older, shorter header: only `size` bytes are written
void smatrix_stats_sz(smatrix_t* self, smatrix_stats_t* out, size_t size) {
  smatrix_stats_t full;
  smatrix_stats(self, &full);
  memcpy(out, &full, std::min(size, sizeof full));
}

void smatrix_profile(smatrix_t* self, int on) {
  Matrix* m = M(self);
  std::lock_guard<std::mutex> g(m->mu);
  get_timing_resolve(m, 0);
  m->profile = on != 0;
  for (int i = 0; i < 4; i++) {
    m->st.kernel_ms[i] = 0;
    m->st.kernel_launches[i] = 0;
    m->st.kernel_ops[i] = 0;
  }
}

static void row_info_locked(Matrix* m, uint32_t x, uint32_t* four) {
  hipStream_t s = m->stream;
  hipLaunchKernelGGL(k_row_info, dim3(1), dim3(1), 0, s, m->d_dir, m->dir_size - 1, m->arena.base, x,
                     m->d_small + 4);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(m->h_small + 4, m->d_small + 4, 16, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  memcpy(four, m->h_small + 4, 16);
}

int smatrix_row_info(smatrix_t* self, uint32_t x, uint32_t* size, uint32_t* used) {
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  uint32_t f[4];
  row_info_locked(m, x, f);
  if (!f[0]) return 0;
  if (size) *size = f[1];
  if (used) *used = f[2];
  return 1;
}

uint32_t smatrix_row_slots(smatrix_t* self, uint32_t x, uint32_t* kv, uint32_t cap_slots) {
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  cache_sync(m, false);
  uint32_t f[4];
  row_info_locked(m, x, f);
  if (!f[0] || !f[3]) return 0;
  uint32_t n = std::min(f[1], cap_slots);
  HIP_OK(hipMemcpy(kv, m->arena.base + (uint64_t)f[3] * UNIT_BYTES, (size_t)n * 8, hipMemcpyDeviceToHost));
  return f[1];
}

// ---- sharding helpers (include/smatrix_shard.h) ------------------------------------------------
uint32_t smatrix_shard_of(uint32_t x, uint32_t nshards) { return shard_of(x, nshards); }

static int partition_impl(size_t n, const uint32_t* d_x, const uint32_t* d_y, const uint32_t* d_v,
                          uint32_t nshards, uint64_t* counts_host, void* d_work, uint32_t* d_perm,
                          uint32_t* d_xo, uint32_t* d_yo, uint32_t* d_vo, uint32_t* d_packed,
                          const uint32_t* d_place, uint32_t place_slots, const uint32_t* d_cuts, void* hip_stream) {
  if (nshards == 0 || nshards > MAX_SHARDS || n >= (1ull << 32)) return -1;
  if (place_slots > PLACE_MAX_SLOTS || (place_slots & (place_slots - 1)) || (place_slots && !d_place)) return -1;
  const uint2* place = reinterpret_cast<const uint2*>(d_place);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  unsigned long long* work = static_cast<unsigned long long*>(d_work);   // 2 x 64 x 8 bytes: counts, cursors
  HIP_OK(hipMemsetAsync(work, 0, 2 * MAX_SHARDS * sizeof(unsigned long long), s));
  if (n) {
    hipLaunchKernelGGL(k_part_count, dim3(std::min<uint32_t>(blocks_for(n), 2048)), dim3(256), 0, s,
                       (uint32_t)n, d_x, nshards, work, place, place_slots, d_cuts);
    hipLaunchKernelGGL(k_part_offsets, dim3(1), dim3(1), 0, s, work, nshards);
    hipLaunchKernelGGL(k_part_scatter, dim3(blocks_for(n, 256 * PART_OPT)), dim3(256), 0, s, (uint32_t)n,
                       d_x, d_y, d_v, nshards, work + MAX_SHARDS, d_perm, d_xo, d_yo, d_vo, d_packed, place, place_slots, d_cuts);
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipMemcpyAsync(counts_host, work, nshards * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int smatrix_partition_dev(size_t n, const uint32_t* d_x, const uint32_t* d_y, const uint32_t* d_v,
                          uint32_t nshards, uint64_t* counts_host, void* d_work, uint32_t* d_perm,
                          uint32_t* d_xo, uint32_t* d_yo, uint32_t* d_vo, const uint32_t* d_place,
                          uint32_t place_slots, const uint32_t* d_cuts, void* hip_stream) {
  return partition_impl(n, d_x, d_y, d_v, nshards, counts_host, d_work, d_perm, d_xo, d_yo, d_vo, nullptr,
                        d_place, place_slots, d_cuts, hip_stream);
}

int smatrix_partition_packed_dev(size_t n, const uint32_t* d_x, const uint32_t* d_y, const uint32_t* d_v,
                                 uint32_t nshards, uint64_t* counts_host, void* d_work, uint32_t* d_perm,
                                 uint32_t* d_packed, const uint32_t* d_place, uint32_t place_slots,
                                 const uint32_t* d_cuts, void* hip_stream) {
  return partition_impl(n, d_x, d_y, d_v, nshards, counts_host, d_work, d_perm, nullptr, nullptr, nullptr,
                        d_packed, d_place, place_slots, d_cuts, hip_stream);
}

uint32_t smatrix_place_slot(uint32_t x, uint32_t slots) { return smx_fmix32(x) & (slots - 1u); }
uint32_t smatrix_shard_mix(uint32_t x) { return shard_mix(x); }

size_t smatrix_displaced_rows(smatrix_t* self, uint32_t rank, uint32_t nshards, uint32_t* out_x, size_t cap) {
  Matrix* m = M(self);
  set_device(m);
  std::lock_guard<std::mutex> g(m->mu);
  uint32_t *d_out = nullptr, *d_cnt = nullptr;
  const uint32_t c = (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu);
  dev_malloc(&d_out, std::max<size_t>(c, 1) * 4);
  dev_malloc(&d_cnt, 4);
  HIP_OK(hipMemsetAsync(d_cnt, 0, 4, m->stream));
  hipLaunchKernelGGL(k_displaced_rows, dim3(std::min<uint32_t>(blocks_for(m->dir_size), 4096)), dim3(256), 0, m->stream,
                     m->d_dir, m->dir_size, rank, nshards, d_out, c, d_cnt);
  HIP_OK(hipGetLastError());
  uint32_t cnt = 0;
  HIP_OK(hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, m->stream));
  HIP_OK(hipStreamSynchronize(m->stream));
  if (out_x && std::min<uint32_t>(cnt, c)) HIP_OK(hipMemcpy(out_x, d_out, (size_t)std::min<uint32_t>(cnt, c) * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_out); (void)hipFree(d_cnt);
  return cnt;
}

int smatrix_unpack_dev(size_t n, uint32_t width, const uint32_t* d_packed, uint32_t* d_x, uint32_t* d_y,
                       uint32_t* d_v, void* hip_stream) {
  if (width != 2 && width != 3) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_unpack, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     (uint32_t)n, width, d_packed, d_x, d_y, d_v);
  HIP_OK(hipGetLastError());
  return 0;
}

int smatrix_gather_dev(size_t n, const uint32_t* d_src, const uint32_t* d_perm, uint32_t* d_out,
                       void* hip_stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_gather, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     (uint32_t)n, d_src, d_perm, d_out);
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- random-access probes (include/smx_probe.h) ------------------------------------------------
int smx_probe_random_dev(void* d_buf, size_t bytes, size_t touches, int mode, uint64_t seed,
                         void* d_sink, void* hip_stream) {
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  uint64_t* buf = static_cast<uint64_t*>(d_buf);
  unsigned long long* sink = static_cast<unsigned long long*>(d_sink);
  const uint64_t words = bytes / 8;
  if (words < 2 || touches == 0) return -1;
  const dim3 grid(256 * 8 * 4), block(256);     // 32 waves per CU on 256 CUs, grid-stride
  switch (mode) {
    case 0: hipLaunchKernelGGL((k_probe_random<0>), grid, block, 0, s, buf, words, (uint64_t)touches, seed, sink); break;
    case 1: hipLaunchKernelGGL((k_probe_random<1>), grid, block, 0, s, buf, words, (uint64_t)touches, seed, sink); break;
    case 2: hipLaunchKernelGGL((k_probe_random<2>), grid, block, 0, s, buf, words, (uint64_t)touches, seed, sink); break;
    case 3: hipLaunchKernelGGL((k_probe_random<3>), grid, block, 0, s, buf, words, (uint64_t)touches, seed, sink); break;
    default: return -1;
  }
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- stream generator, device side (include/smx_stream.h) -------------------------------
int smx_stream_fill_device(smx_stream_t* st, uint64_t first, size_t n, uint32_t* d_x, uint32_t* d_y,
                           void* hip_stream) {
  if (n == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  if (st->dist == SMX_DIST_ZIPF && !st->d_cdf) {
    dev_malloc(&st->d_cdf, (size_t)st->n_ids * sizeof(double));
    HIP_OK(hipMemcpy(st->d_cdf, st->cdf, (size_t)st->n_ids * sizeof(double), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(k_stream_fill, dim3(blocks_for(n)), dim3(256), 0, s, st->dist, st->seed,
                     st->n_ids, st->d_cdf, st->scramble, first, (uint64_t)n, d_x, d_y,
                     st->dist == SMX_DIST_CF ? (uint64_t)st->zipf_s : 1ull);
  HIP_OK(hipGetLastError());
  return 0;
}

void smx_stream_release_device(smx_stream_t* st) {
  if (st->d_cdf) (void)hipFree(st->d_cdf);
  st->d_cdf = nullptr;
}

}  // extern "C"

// ---- the multi-GPU router (include/smatrix_shard.h) ---------------------------------------------------
#include "smx_shard.inc"
