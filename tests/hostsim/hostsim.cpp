// Runtime of the kernel host simulation (see hip/hip_runtime.h in this directory for what it
// is for).  Test infrastructure only.
//
//  * memory: every hipMalloc is its own mmap with PROT_NONE pages on both sides.  The buffer
//    is rounded up to 256 bytes and sits flush against the rear guard (HOSTSIM_GUARD=end, the
//    default) or against the front guard (HOSTSIM_GUARD=start), so an access outside it
//    raises SIGSEGV at once; the handler prints the kernel, workgroup, thread and the nearest
//    allocation, then aborts.  (A real GPU only faults at page granularity; this is stricter.)
//  * dynamic LDS: one guarded region per worker thread, the launch's byte count flush against
//    the rear guard.
//  * threads of a workgroup: fibers switched by a few instructions of assembly; a fiber runs
//    until its next cross-lane operation or barrier.  Workgroups of a launch are spread over
//    HOSTSIM_THREADS worker threads (default: the machine's cores, at most 8).  HOSTSIM_ORDER =
//    reverse | shuffle:<seed> changes the order in which the scheduler visits the threads.
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

// HOSTSIM_TSAN: the kernels are compiled with -fsanitize=thread and every simulated thread is a
// fiber of its own to the race detector.  What orders two threads is told to it explicitly:
// workgroup barriers, cross-lane operations of a wave, the start and the end of a workgroup.
// Switching fibers orders nothing, so two threads that touch the same LDS or device memory
// (one of them writing, not both atomically) with none of these in between are reported, however
// the simulation happened to interleave them.  This file itself is built without
// instrumentation: the scheduler's own state is not what is being checked.
#ifdef HOSTSIM_TSAN
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
void __tsan_acquire(void* addr);
void __tsan_release(void* addr);
}
#define TSAN_ACQUIRE(p) __tsan_acquire((void*)(p))
#define TSAN_RELEASE(p) __tsan_release((void*)(p))
#define TSAN_SWITCH(f) __tsan_switch_to_fiber((f), 1u /* no synchronisation */)
#else
#define TSAN_ACQUIRE(p) ((void)0)
#define TSAN_RELEASE(p) ((void)0)
#define TSAN_SWITCH(f) ((void)0)
#endif

namespace hostsim {

static const size_t kPage = 4096;
static const size_t kStack = 256 * 1024;
static const int kMaxThreads = 1024;
static const size_t kLdsMax = 160 * 1024;

// HOSTSIM_FILL=<byte>: what fresh device memory and LDS hold (default 0xA5 / 0xCD); results that
// change with it depend on memory nobody wrote
static int fill_byte(int dflt) {
  static int v = [] { const char* e = getenv("HOSTSIM_FILL"); return e ? (int)strtol(e, nullptr, 0) & 0xFF : -1; }();
  return v >= 0 ? v : dflt;
}


// ------------------------------------------------------------------------------------------
// fibers
// ------------------------------------------------------------------------------------------
extern "C" void hostsim_switch(void** saveSp, void* loadSp);
asm(R"(
.text
.globl hostsim_switch
.type hostsim_switch,@function
hostsim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hostsim_switch,.-hostsim_switch
)");

struct Wave {
  int alive = 0, arrived = 0;
  uint64_t aliveMask = 0;
  volatile uint64_t gen = 0;
  int site = 0;
  uint64_t slot[2][64];
  uint64_t maskAt[2];
  uint64_t order = 0;   // what the race detector's view of cross-lane operations hangs on
};

struct Block {
  const char* kernel = "";
  uint3 bid;
  dim3 bdim, gdim;
  int nthreads = 0, nwaves = 0;
  int alive = 0, arrived = 0;
  volatile uint64_t gen = 0;
  int site = 0;
  int orAcc[2] = {0, 0};
  unsigned barCount = 0;
  uint8_t* dynLds = nullptr;
  const std::function<void()>* fn = nullptr;
  void* schedSp = nullptr;
  Lane lanes[kMaxThreads];
  Wave waves[kMaxThreads / 64];
  uint8_t* stacks = nullptr;     // kMaxThreads * kStack
  uint8_t* ldsRegion = nullptr;  // kLdsMax + guard
  // race detector: one fiber per thread slot, kept for the life of this object (a slot's
  // successive threads are ordered by that, which also covers their reuse of the slot's stack)
  void* laneFiber[kMaxThreads] = {};
  void* schedFiber = nullptr;
  uint64_t orderStart = 0, orderBarrier = 0, orderDone = 0;
};

thread_local Lane* tl_lane = nullptr;
thread_local Block* tl_block = nullptr;

const uint3& thread_idx() { return tl_lane->tid; }
const uint3& block_idx() { return tl_block->bid; }
const dim3& block_dim() { return tl_block->bdim; }
const dim3& grid_dim() { return tl_block->gdim; }
int lane_in_wave() { return tl_lane->lin & 63; }
uint8_t* dyn_lds() { return tl_block->dynLds; }

static void describe_position(FILE* f) {
  Block* b = tl_block;
  Lane* l = tl_lane;
  if (b && l)
    fprintf(f, "  kernel %s  workgroup (%u,%u,%u) of (%u,%u,%u)  thread (%u,%u,%u)\n", b->kernel, b->bid.x,
            b->bid.y, b->bid.z, b->gdim.x, b->gdim.y, b->gdim.z, l->tid.x, l->tid.y, l->tid.z);
  else
    fprintf(f, "  (host code, no kernel running on this thread)\n");
}

[[noreturn]] void fail(const char* what, int site) {
  fprintf(stderr, "hostsim: %s (source line %d)\n", what, site);
  describe_position(stderr);
  fflush(stderr);
  abort();
}

static inline void yield_to_scheduler() {
  Lane* l = tl_lane;
  TSAN_SWITCH(tl_block->schedFiber);
  hostsim_switch(&l->sp, tl_block->schedSp);
}

static void wait_for(const volatile uint64_t* gen, uint64_t seen) {
  Lane* l = tl_lane;
  l->waitOn = gen;
  l->waitVal = seen;
  while (*gen == seen) yield_to_scheduler();
  l->waitOn = nullptr;
}

Gathered wave_gather(int site, uint64_t mine) {
  Block* b = tl_block;
  Lane* l = tl_lane;
  Wave& w = b->waves[l->lin >> 6];
  const int par = (int)(l->opCount++ & 1u);
  const int lane = l->lin & 63;
  TSAN_RELEASE(&w.order);
  if (w.arrived == 0) w.site = site;
  else if (w.site != site) {
    fprintf(stderr, "hostsim: lanes of one wave meet at different cross-lane operations (lines %d and %d)\n",
            w.site, site);
    fail("non-convergent wave-level code", site);
  }
  w.slot[par][lane] = mine;
  ++w.arrived;
  if (w.arrived == w.alive) {
    w.maskAt[par] = w.aliveMask;
    w.arrived = 0;
    w.gen = w.gen + 1;
  } else {
    l->site = site;
    wait_for(&w.gen, w.gen);
  }
  TSAN_ACQUIRE(&w.order);
  return Gathered{w.slot[par], w.maskAt[par]};
}

static void barrier_impl(int site, int pred, int* orOut) {
  Block* b = tl_block;
  Lane* l = tl_lane;
  TSAN_RELEASE(&b->orderBarrier);
  if (b->arrived == 0) b->site = site;
  else if (b->site != site) {
    fprintf(stderr, "hostsim: threads of one workgroup meet at different barriers (lines %d and %d)\n", b->site,
            site);
    fail("non-convergent barrier", site);
  }
  const int par = (int)(b->barCount & 1u);
  if (pred) b->orAcc[par] = 1;
  ++b->arrived;
  if (b->arrived == b->alive) {
    b->arrived = 0;
    b->orAcc[par ^ 1] = 0;   // the next barrier's accumulator
    b->barCount++;
    b->gen = b->gen + 1;
  } else {
    l->site = site;
    wait_for(&b->gen, b->gen);
  }
  TSAN_ACQUIRE(&b->orderBarrier);
  if (orOut) *orOut = b->orAcc[par];
}

void block_barrier(int site) { barrier_impl(site, 0, nullptr); }
int block_barrier_or(int site, int pred) {
  int r = 0;
  barrier_impl(site, pred, &r);
  return r;
}

// a thread has returned from the kernel
static void lane_exit() {
  Block* b = tl_block;
  Lane* l = tl_lane;
  Wave& w = b->waves[l->lin >> 6];
  // a thread that has returned no longer takes part in barriers; what it wrote before is
  // treated as visible to those who pass the next one
  TSAN_RELEASE(&w.order);
  TSAN_RELEASE(&b->orderBarrier);
  TSAN_RELEASE(&b->orderDone);
  l->done = true;
  --w.alive;
  w.aliveMask &= ~(1ull << (l->lin & 63));
  --b->alive;
  // the rest of the wave / workgroup may have been waiting for this thread only
  if (w.alive > 0 && w.arrived == w.alive) {
    // values were deposited with the parity of the waiting lanes' last operation
    // (all waiting lanes share it); find it from any waiting lane
    for (int i = 0; i < 64; ++i) {
      Lane& o = b->lanes[(l->lin & ~63) + i];
      if ((l->lin & ~63) + i < b->nthreads && !o.done && o.waitOn == &w.gen) {
        w.maskAt[(o.opCount - 1) & 1u] = w.aliveMask;
        break;
      }
    }
    w.arrived = 0;
    w.gen = w.gen + 1;
  }
  if (b->alive > 0 && b->arrived == b->alive) {
    const int par = (int)(b->barCount & 1u);
    b->arrived = 0;
    b->orAcc[par ^ 1] = 0;
    b->barCount++;
    b->gen = b->gen + 1;
  }
}

static void fiber_main() {
  TSAN_ACQUIRE(&tl_block->orderStart);
  (*tl_block->fn)();
  lane_exit();
  for (;;) yield_to_scheduler();
}

static void prepare_fiber(Block* b, int i) {
  uint8_t* top = b->stacks + (size_t)(i + 1) * kStack;
  void** sp = (void**)top;
  // layout consumed by hostsim_switch: r15 r14 r13 r12 rbx rbp, return address; after the
  // `ret` the stack pointer must be 8 mod 16, as right after a call
  *--sp = nullptr;               // padding so that the entry sees a call-like alignment
  *--sp = (void*)&fiber_main;    // return address
  for (int k = 0; k < 6; ++k) *--sp = nullptr;
  b->lanes[i].sp = sp;
}

static std::mutex g_blockMu;
// idle workgroup states; released when the process ends so that a leak check of the library's
// host code reports only what the library leaks
static struct BlockPool : std::vector<Block*> {
  ~BlockPool();
} g_freeBlocks;

// per-worker state (fiber stacks, LDS region) is kept in a free list between launches
static std::mutex g_allocMu;
struct Alloc { uint8_t* map; size_t mapBytes; size_t bytes; };
static std::map<uintptr_t, Alloc> g_allocs;   // user pointer -> mapping

static Block* acquire_block() {
  {
    std::lock_guard<std::mutex> lk(g_blockMu);
    if (!g_freeBlocks.empty()) {
      Block* b = g_freeBlocks.back();
      g_freeBlocks.pop_back();
      return b;
    }
  }
  Block* blk = new Block();
  // stacks with a guard page below each one
  blk->stacks = (uint8_t*)mmap(nullptr, (size_t)kMaxThreads * kStack, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (blk->stacks == MAP_FAILED) { perror("hostsim: mmap stacks"); abort(); }
  for (int i = 0; i < kMaxThreads; ++i) mprotect(blk->stacks + (size_t)i * kStack, kPage, PROT_NONE);
  blk->ldsRegion = (uint8_t*)mmap(nullptr, kLdsMax + 2 * kPage, PROT_READ | PROT_WRITE,
                                  MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (blk->ldsRegion == MAP_FAILED) { perror("hostsim: mmap lds"); abort(); }
  mprotect(blk->ldsRegion, kPage, PROT_NONE);
  mprotect(blk->ldsRegion + kPage + kLdsMax, kPage, PROT_NONE);
  return blk;
}

BlockPool::~BlockPool() {
  for (Block* b : *this) {
    if (b->stacks) munmap(b->stacks, (size_t)kMaxThreads * kStack);
    if (b->ldsRegion) munmap(b->ldsRegion, kLdsMax + 2 * kPage);
    delete b;
  }
}

static void release_block(Block* b) {
  std::lock_guard<std::mutex> lk(g_blockMu);
  g_freeBlocks.push_back(b);
}

static void run_block(Block* b, const char* name, uint3 bid, dim3 grid, dim3 block, size_t ldsBytes,
                      const std::function<void()>* fn) {
  const int nt = (int)(block.x * block.y * block.z);
  b->kernel = name;
  b->bid = bid;
  b->bdim = block;
  b->gdim = grid;
  b->nthreads = nt;
  b->nwaves = (nt + 63) / 64;
  b->alive = nt;
  b->arrived = 0;
  b->orAcc[0] = b->orAcc[1] = 0;
  b->barCount = 0;
  b->fn = fn;
  const size_t ldsRounded = (ldsBytes + 15) & ~(size_t)15;
  b->dynLds = b->ldsRegion + kPage + kLdsMax - ldsRounded;
  memset(b->dynLds, fill_byte(0xCD), ldsRounded);   // LDS starts out undefined
  for (int w = 0; w < b->nwaves; ++w) {
    Wave& wv = b->waves[w];
    const int n = std::min(64, nt - 64 * w);
    wv.alive = n;
    wv.arrived = 0;
    wv.aliveMask = n == 64 ? ~0ull : ((1ull << n) - 1ull);
  }
  for (int i = 0; i < nt; ++i) {
    Lane& l = b->lanes[i];
    l.lin = i;
    l.tid.x = (unsigned)i % block.x;
    l.tid.y = ((unsigned)i / block.x) % block.y;
    l.tid.z = (unsigned)i / (block.x * block.y);
    l.done = false;
    l.waitOn = nullptr;
    l.opCount = 0;
    l.site = 0;
    prepare_fiber(b, i);
  }
  tl_block = b;
#ifdef HOSTSIM_TSAN
  b->schedFiber = __tsan_get_current_fiber();
  for (int i = 0; i < nt; ++i)
    if (!b->laneFiber[i]) b->laneFiber[i] = __tsan_create_fiber(0);
#endif
  TSAN_RELEASE(&b->orderStart);
  // HOSTSIM_ORDER=reverse runs the threads of a workgroup from the last to the first: code whose
  // result depends on which thread gets to a point first (a missing barrier) changes its output
  static const bool reverseOrder = [] { const char* e = getenv("HOSTSIM_ORDER"); return e && !strcmp(e, "reverse"); }();
  // HOSTSIM_ORDER=shuffle:<seed>: a fresh pseudo-random order of the waves, and of the lanes
  // inside each wave, for every sweep of the scheduler
  static const long shuffleSeed = [] {
    const char* e = getenv("HOSTSIM_ORDER");
    return (e && !strncmp(e, "shuffle:", 8)) ? atol(e + 8) + 1 : 0L;
  }();
  uint64_t rs = (uint64_t)shuffleSeed * 0x9E3779B97F4A7C15ull + bid.x * 1315423911ull + bid.y * 2654435761ull;
  auto rnd = [&rs]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
  std::vector<int> order;
  if (shuffleSeed) order.resize(nt);
  while (b->alive > 0) {
    bool progressed = false;
    if (shuffleSeed) {
      std::vector<int> waves(b->nwaves);
      for (int w = 0; w < b->nwaves; ++w) waves[w] = w;
      for (int w = b->nwaves - 1; w > 0; --w) std::swap(waves[w], waves[rnd() % (w + 1)]);
      int at = 0;
      for (int w : waves) {
        const int first = at, cnt = std::min(64, nt - 64 * w);
        for (int l = 0; l < cnt; ++l) order[at++] = 64 * w + l;
        for (int l = cnt - 1; l > 0; --l) std::swap(order[first + l], order[first + rnd() % (l + 1)]);
      }
    }
    for (int k = 0; k < nt; ++k) {
      const int i = shuffleSeed ? order[k] : (reverseOrder ? nt - 1 - k : k);
      Lane& l = b->lanes[i];
      if (l.done) continue;
      if (l.waitOn && *l.waitOn == l.waitVal) continue;
      tl_lane = &l;
      TSAN_SWITCH(b->laneFiber[i]);
      hostsim_switch(&b->schedSp, l.sp);
      progressed = true;
    }
    if (!progressed) {
      fprintf(stderr, "hostsim: deadlock in kernel %s, workgroup (%u,%u,%u): every thread waits\n", name, bid.x,
              bid.y, bid.z);
      for (int i = 0; i < nt; ++i) {
        Lane& l = b->lanes[i];
        if (!l.done)
          fprintf(stderr, "  thread %d waits at source line %d (%s)\n", i, l.site,
                  l.waitOn == &b->gen ? "workgroup barrier" : "cross-lane operation");
      }
      abort();
    }
  }
  tl_lane = nullptr;
  TSAN_ACQUIRE(&b->orderDone);
}

static int worker_count() {
  static int n = [] {
    const char* e = getenv("HOSTSIM_THREADS");
    int v = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    return std::max(1, std::min(v, 64));
  }();
  return n;
}

static std::atomic<long> g_launches{0};

// dynamic LDS a kernel may be launched with: 64 KB unless hipFuncSetAttribute raised it
static std::mutex g_attrMu;
static std::map<const void*, size_t> g_maxDynLds;

void launch(const char* name, const void* kernel, dim3 grid, dim3 block, size_t ldsBytes,
            const std::function<void()>& fn) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
  const int nt = (int)(block.x * block.y * block.z);
  if (grid.y > 65535u || grid.z > 65535u) {
    fprintf(stderr, "hostsim: %s: grid (%u,%u,%u) exceeds the 65535 limit of the y / z dimensions\n", name, grid.x,
            grid.y, grid.z);
    abort();
  }
  {
    std::lock_guard<std::mutex> lk(g_attrMu);
    auto it = g_maxDynLds.find(kernel);
    const size_t allowed = it == g_maxDynLds.end() ? 64 * 1024 : std::max<size_t>(64 * 1024, it->second);
    if (ldsBytes > allowed) {
      fprintf(stderr, "hostsim: %s: %zu bytes of dynamic LDS without hipFuncSetAttribute (allowed %zu)\n", name,
              ldsBytes, allowed);
      abort();
    }
  }
  if (nblocks <= 0 || nt <= 0) { fprintf(stderr, "hostsim: empty launch of %s\n", name); abort(); }
  if (nt > kMaxThreads) { fprintf(stderr, "hostsim: %s: %d threads per workgroup\n", name, nt); abort(); }
  if (ldsBytes > kLdsMax) { fprintf(stderr, "hostsim: %s: %zu bytes of dynamic LDS\n", name, ldsBytes); abort(); }
  g_launches++;
  if (getenv("HOSTSIM_TRACE"))
    fprintf(stderr, "hostsim: launch %s grid (%u,%u,%u) block (%u,%u,%u) lds %zu\n", name, grid.x, grid.y, grid.z,
            block.x, block.y, block.z, ldsBytes);
  std::atomic<long> next{0};
  auto work = [&]() {
    Block* b = acquire_block();
    for (;;) {
      const long i = next.fetch_add(1);
      if (i >= nblocks) break;
      uint3 bid;
      bid.x = (unsigned)(i % grid.x);
      bid.y = (unsigned)((i / grid.x) % grid.y);
      bid.z = (unsigned)(i / ((long)grid.x * grid.y));
      run_block(b, name, bid, grid, block, ldsBytes, &fn);
    }
    tl_block = nullptr;
    release_block(b);
  };
  auto probe = [&](const char* when) {
    if (!getenv("HOSTSIM_PROBE")) return;
    std::lock_guard<std::mutex> lk(g_allocMu);
    for (auto& kv : g_allocs) {
      unsigned char vec[1];
      const uintptr_t pg = kv.first & ~(uintptr_t)(kPage - 1);
      if (mincore((void*)pg, kPage, vec) != 0)
        fprintf(stderr, "hostsim: %s %s: allocation %p (%zu bytes) is not mapped\n", when, name, (void*)kv.first,
                kv.second.bytes);
    }
  };
  probe("before");
  const int nw = (int)std::min<long>(worker_count(), nblocks);
  if (nw <= 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 1; t < nw; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  probe("after");
}

// ------------------------------------------------------------------------------------------
// guarded allocations
// ------------------------------------------------------------------------------------------
static bool guard_at_start() {
  static int v = [] { const char* e = getenv("HOSTSIM_GUARD"); return e && !strcmp(e, "start") ? 1 : 0; }();
  return v != 0;
}

static void* guarded_alloc(size_t n) {
  const size_t body = std::max<size_t>(256, (n + 255) & ~(size_t)255);
  const size_t bodyPages = (body + kPage - 1) / kPage * kPage;
  const size_t total = bodyPages + 2 * kPage;
  uint8_t* m = (uint8_t*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return nullptr;
  mprotect(m, kPage, PROT_NONE);
  mprotect(m + kPage + bodyPages, kPage, PROT_NONE);
  uint8_t* user = guard_at_start() ? m + kPage : m + kPage + bodyPages - body;
  memset(m + kPage, fill_byte(0xA5), bodyPages);   // device memory starts out undefined
  std::lock_guard<std::mutex> lk(g_allocMu);
  g_allocs[(uintptr_t)user] = Alloc{m, total, n};
  return user;
}

static bool guarded_free(void* p) {
  std::lock_guard<std::mutex> lk(g_allocMu);
  auto it = g_allocs.find((uintptr_t)p);
  if (it == g_allocs.end()) return false;
  munmap(it->second.map, it->second.mapBytes);
  g_allocs.erase(it);
  return true;
}

static void on_segv(int, siginfo_t* si, void*) {
  const uintptr_t a = (uintptr_t)si->si_addr;
  fprintf(stderr, "hostsim: invalid memory access at %p\n", si->si_addr);
  describe_position(stderr);
  // nearest allocation (no lock: the process is about to abort)
  const Alloc* best = nullptr;
  uintptr_t bestUser = 0;
  for (auto& kv : g_allocs) {
    const uintptr_t lo = (uintptr_t)kv.second.map, hi = lo + kv.second.mapBytes;
    if (a >= lo && a < hi) { best = &kv.second; bestUser = kv.first; }
  }
  if (best)
    fprintf(stderr, "  the address is %ld bytes from the start of a %zu-byte device allocation (%s it)\n",
            (long)(a - bestUser), best->bytes, a < bestUser ? "before" : "past the end of");
  else if (tl_block) {
    const uintptr_t l0 = (uintptr_t)tl_block->ldsRegion, l1 = l0 + kLdsMax + 2 * kPage;
    if (a >= l0 && a < l1)
      fprintf(stderr, "  the address is %ld bytes from the start of the workgroup's dynamic LDS\n",
              (long)(a - (uintptr_t)tl_block->dynLds));
    else {
      const uintptr_t s0 = (uintptr_t)tl_block->stacks, s1 = s0 + (size_t)kMaxThreads * kStack;
      if (a >= s0 && a < s1) fprintf(stderr, "  the address is in a fiber stack guard (stack overflow)\n");
      else fprintf(stderr, "  the address belongs to no device allocation\n");
    }
  }
  if (FILE* mf = fopen("/proc/self/maps", "r")) {
    char line[512];
    while (fgets(line, sizeof(line), mf)) {
      unsigned long lo = 0, hi = 0;
      if (sscanf(line, "%lx-%lx", &lo, &hi) == 2 && a >= lo - 8192 && a < hi + 8192) fprintf(stderr, "  maps: %s", line);
    }
    fclose(mf);
  }
  {
    void* frames[48];
    const int nf = backtrace(frames, 48);
    backtrace_symbols_fd(frames, nf, 2);
  }
  fflush(stderr);
  signal(SIGSEGV, SIG_DFL);
  abort();
}

static void install_handler() {
  static std::once_flag once;
  std::call_once(once, [] {
    static uint8_t altstack[64 * 1024];
    stack_t ss;
    ss.ss_sp = altstack;
    ss.ss_size = sizeof(altstack);
    ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
  });
}

}  // namespace hostsim

// ------------------------------------------------------------------------------------------
// runtime API subset: everything is synchronous
// ------------------------------------------------------------------------------------------
struct hostsim_stream { int dummy; };
struct hostsim_event { std::chrono::steady_clock::time_point t; bool recorded = false; };

extern "C" {

hipError_t hipMalloc(void** p, size_t n) {
  hostsim::install_handler();
  if (!p) return hipErrorInvalidValue;
  *p = hostsim::guarded_alloc(n);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  return hostsim::guarded_free(p) ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t* pool, int) { *pool = nullptr; return hipErrorInvalidValue; }
hipError_t hipMemPoolSetAttribute(hipMemPool_t, hipMemPoolAttr, void*) { return hipSuccess; }
hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
hipError_t hipFreeAsync(void* p, hipStream_t) { return hipFree(p); }
// page-locked host allocations are remembered so that hipPointerGetAttributes can tell them
// from ordinary host memory, as the runtime does
static std::mutex g_pinned_mu;
static std::map<uintptr_t, size_t> g_pinned;
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
  const hipError_t e = hipMalloc(p, n);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned[(uintptr_t)*p] = n;
  }
  return e;
}
hipError_t hipHostFree(void* p) {
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned.erase((uintptr_t)p);
  }
  return hipFree(p);
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  auto it = g_pinned.upper_bound((uintptr_t)p);
  if (it != g_pinned.begin()) {
    --it;
    if ((uintptr_t)p < it->first + it->second) { a->type = hipMemoryTypeHost; return hipSuccess; }
  }
  return hipErrorInvalidValue;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  if (getenv("HOSTSIM_TRACE")) fprintf(stderr, "hostsim: memcpy %p <- %p, %zu bytes\n", d, s, n);
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
  return hipMemcpy(d, s, n, k);
}
hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, hipMemcpyKind) {
  for (size_t y = 0; y < h; ++y) memmove((uint8_t*)d + y * dpitch, (const uint8_t*)s + y * spitch, w);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h,
                            hipMemcpyKind k, hipStream_t) {
  return hipMemcpy2D(d, dpitch, s, spitch, w, h, k);
}
hipError_t hipMemset(void* d, int v, size_t n) {
  if (n) memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hostsim error"; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hostsim_stream(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostsim_event(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::steady_clock::now();
  e->recorded = true;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  if (!a->recorded || !b->recorded) return hipErrorInvalidValue;
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v) {
  if (a == hipFuncAttributeMaxDynamicSharedMemorySize) {
    if (v < 0 || (size_t)v > hostsim::kLdsMax) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(hostsim::g_attrMu);
    hostsim::g_maxDynLds[f] = (size_t)v;
  }
  return hipSuccess;
}

}  // extern "C"
