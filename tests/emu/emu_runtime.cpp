// Fiber scheduler of the SIMT emulator (tests/emu/hip/hip_runtime.h).  TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>

namespace emu {
thread_local Block* blk = nullptr;
static const size_t STACK = 1u << 20;

struct Tramp { void (*fn)(void*); void* arg; };
static thread_local Tramp g_tramp;

static void lane_main() {
  Block* b = blk;
  g_tramp.fn(g_tramp.arg);
  Lane& me = b->lanes[b->cur];
  me.done = true;
  b->live--;
  const int q = me.tid >> 2;
  b->qlive[q]--;
  // a lane that leaves may be the one the others are waiting for
  if (b->live > 0 && b->arrived >= b->live) { b->arrived = 0; b->generation++; }
  if (b->qlive[q] > 0 && b->qarrived[q] >= b->qlive[q]) { b->qarrived[q] = 0; b->qgen[q]++; }
  const int w = me.tid >> 6;
  b->wlive[w]--;
  if (b->wlive[w] > 0 && b->warrived[w] >= b->wlive[w]) { b->warrived[w] = 0; b->wgen[w]++; }
  swapcontext(&me.ctx, &b->sched);
}

void run_block(int bid, int nthreads, void (*fn)(void*), void* arg) {
  static thread_local Block* cache = nullptr;
  if (!cache) {
    cache = new Block();
    cache->lanes.resize(1024);
    for (auto& l : cache->lanes) l.stack = nullptr;
  }
  Block* b = cache;
  blk = b;
  b->nthreads = nthreads; b->bid = bid; b->live = nthreads; b->arrived = 0; b->generation = 0;
  for (int w = 0; w < (nthreads + 63) / 64; w++) { b->wlive[w] = (nthreads - 64 * w) < 64 ? (nthreads - 64 * w) : 64; b->warrived[w] = 0; b->wgen[w] = 0; }
  for (int q = 0; q < (nthreads + 3) / 4; q++) { b->qlive[q] = (nthreads - 4 * q) < 4 ? (nthreads - 4 * q) : 4; b->qarrived[q] = 0; b->qgen[q] = 0; }
  g_tramp = Tramp{fn, arg};
  for (int t = 0; t < nthreads; t++) {
    Lane& l = b->lanes[t];
    if (!l.stack) l.stack = (char*)malloc(STACK);
    l.tid = t; l.done = false;
    getcontext(&l.ctx);
    l.ctx.uc_stack.ss_sp = l.stack;
    l.ctx.uc_stack.ss_size = STACK;
    l.ctx.uc_link = &b->sched;
    makecontext(&l.ctx, (void (*)())lane_main, 0);
  }
  while (b->live > 0)
    for (int t = 0; t < nthreads; t++) {
      if (b->lanes[t].done) continue;
      b->cur = t;
      swapcontext(&b->sched, &b->lanes[t].ctx);
    }
  blk = nullptr;
}
}  // namespace emu
