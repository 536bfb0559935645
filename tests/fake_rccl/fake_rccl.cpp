// TEST INFRASTRUCTURE: an in-process stand-in for the dozen RCCL entry points libgfft's exchange
// module binds (csrc/exchange.cpp), so that the native wire -- communicator split, grouped
// send/recv, stream ordering -- can run on ONE GPU with thread-ranks (tests/thread_comm.py).  RCCL
// itself refuses two ranks on one device.  Every "rank" is a thread of the test process; a message
// is a device-to-device copy enqueued on the RECEIVER's stream after an event the sender recorded,
// and the sender's stream waits for the copy before it may reuse the buffer -- the ordering
// contract of ncclSend / ncclRecv.  Loaded through gfft_rccl_load(path); never shipped.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg { const void *ptr; size_t bytes; hipEvent_t ready; };

struct World {
  int size;
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Msg>> box;          // (src, dst) -> posted sends
  std::map<std::pair<int, int>, std::deque<hipEvent_t>> acks;  // (src, dst) -> "copied" events
  // split rendezvous
  int gen = 0, arrived = 0;
  std::vector<std::pair<int, int>> colorkey;                   // per rank, current generation
  std::map<std::pair<int, int>, std::shared_ptr<World>> children;  // (gen, color)
  std::map<int, std::vector<std::pair<int, int>>> snapshot;    // gen -> colorkey table
  explicit World(int n) : size(n), colorkey(n) {}
};

struct Comm { std::shared_ptr<World> w; int rank; };

struct Op { bool send; void *ptr; size_t bytes; int peer; Comm *c; hipStream_t s; };

std::mutex g_reg_m;
std::condition_variable g_reg_cv;
std::map<std::string, std::pair<std::shared_ptr<World>, int>> g_registry;   // id -> (world, joined)
int g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

// FAKE_RCCL_DELAY_US=<n>: every message lands n microseconds late (a spin kernel on the receiver's
// stream ahead of the copy), so a consumer that does not wait for the arrival event reads stale
// data instead of getting away with it because device copies of test-sized messages are instant.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

long long delay_ticks() {
  static const long long t = [] {
    const char *e = getenv("FAKE_RCCL_DELAY_US");
    return e ? atoll(e) * 100 : 0LL;          // wall_clock64 runs at 100 MHz
  }();
  return t;
}

size_t dtype_size(int dt) {
  switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; }
  return 1;
}

int run_group(std::vector<Op> &ops) {
  // 1. post the sends
  std::map<hipStream_t, hipEvent_t> ready;
  for (Op &o : ops) {
    if (!o.send) continue;
    if (!ready.count(o.s)) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 1;
      if (hipEventRecord(e, o.s) != hipSuccess) return 1;
      ready[o.s] = e;
    }
    World &w = *o.c->w;
    std::lock_guard<std::mutex> lk(w.m);
    w.box[{o.c->rank, o.peer}].push_back(Msg{o.ptr, o.bytes, ready[o.s]});
    w.cv.notify_all();
  }
  // 2. receive: copy on my stream once the sender's data is ready
  for (Op &o : ops) {
    if (o.send) continue;
    World &w = *o.c->w;
    Msg msg;
    {
      std::unique_lock<std::mutex> lk(w.m);
      auto key = std::make_pair(o.peer, o.c->rank);
      w.cv.wait(lk, [&] { return !w.box[key].empty(); });
      msg = w.box[key].front();
      w.box[key].pop_front();
    }
    if (msg.bytes != o.bytes) return 5;     // ncclInvalidArgument: mismatched message sizes
    if (hipStreamWaitEvent(o.s, msg.ready, 0) != hipSuccess) return 1;
    if (delay_ticks() > 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, o.s, delay_ticks());
    if (o.bytes && hipMemcpyAsync(o.ptr, msg.ptr, o.bytes, hipMemcpyDeviceToDevice, o.s) != hipSuccess) return 1;
    hipEvent_t done;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(done, o.s) != hipSuccess) return 1;
    std::lock_guard<std::mutex> lk(w.m);
    w.acks[{o.peer, o.c->rank}].push_back(done);
    w.cv.notify_all();
  }
  // 3. a send buffer may be reused once the receiver has copied it
  for (Op &o : ops) {
    if (!o.send) continue;
    World &w = *o.c->w;
    hipEvent_t done;
    {
      std::unique_lock<std::mutex> lk(w.m);
      auto key = std::make_pair(o.c->rank, o.peer);
      w.cv.wait(lk, [&] { return !w.acks[key].empty(); });
      done = w.acks[key].front();
      w.acks[key].pop_front();
    }
    if (hipStreamWaitEvent(o.s, done, 0) != hipSuccess) return 1;
  }
  return 0;
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetVersion(int *v) { *v = 0; return 0; }          // 0: "not a real RCCL"
const char *ncclGetErrorString(int r) { return r == 0 ? "success" : r == 5 ? "invalid argument (fake rccl)" : "error (fake rccl)"; }

int ncclGetUniqueId(ncclUniqueId *id) {
  std::lock_guard<std::mutex> lk(g_reg_m);
  memset(id->internal, 0, sizeof id->internal);
  snprintf(id->internal, sizeof id->internal, "fake-rccl-%d", g_next_id++);
  return 0;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
  std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
  std::unique_lock<std::mutex> lk(g_reg_m);
  auto &slot = g_registry[key];
  if (!slot.first) slot.first = std::make_shared<World>(nranks);
  if (slot.first->size != nranks) return 5;
  std::shared_ptr<World> w = slot.first;
  ++slot.second;
  g_reg_cv.notify_all();
  g_reg_cv.wait(lk, [&] { return g_registry[key].second >= nranks; });   // ncclCommInitRank is collective
  Comm *c = new Comm{w, rank};
  *comm = c;
  return 0;
}

int ncclCommSplit(void *comm, int color, int key, void **newcomm, void *) {
  Comm *c = static_cast<Comm *>(comm);
  World &w = *c->w;
  std::unique_lock<std::mutex> lk(w.m);
  const int gen = w.gen;
  w.colorkey[c->rank] = {color, key};
  if (++w.arrived == w.size) {
    w.snapshot[gen] = w.colorkey;
    w.arrived = 0;
    ++w.gen;
    w.cv.notify_all();
  } else {
    w.cv.wait(lk, [&] { return w.gen > gen; });
  }
  *newcomm = nullptr;
  if (color < 0) return 0;
  const auto &table = w.snapshot[gen];
  std::vector<std::pair<int, int>> members;            // (key, parent rank)
  for (int r = 0; r < w.size; ++r)
    if (table[r].first == color) members.push_back({table[r].second, r});
  std::sort(members.begin(), members.end());
  int myrank = 0;
  for (size_t i = 0; i < members.size(); ++i)
    if (members[i].second == c->rank) myrank = (int)i;
  auto &child = w.children[{gen, color}];
  if (!child) child = std::make_shared<World>((int)members.size());
  *newcomm = new Comm{child, myrank};
  return 0;
}

int ncclCommDestroy(void *comm) { delete static_cast<Comm *>(comm); return 0; }
int ncclCommCount(void *comm, int *n) { *n = static_cast<Comm *>(comm)->w->size; return 0; }
int ncclCommUserRank(void *comm, int *r) { *r = static_cast<Comm *>(comm)->rank; return 0; }

int ncclGroupStart() { ++t_depth; return 0; }
int ncclGroupEnd() {
  if (--t_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_group(ops);
}

int ncclSend(const void *buf, size_t count, int dt, int peer, void *comm, hipStream_t s) {
  t_ops.push_back(Op{true, const_cast<void *>(buf), count * dtype_size(dt), peer, static_cast<Comm *>(comm), s});
  if (t_depth == 0) { std::vector<Op> ops; ops.swap(t_ops); return run_group(ops); }
  return 0;
}

int ncclRecv(void *buf, size_t count, int dt, int peer, void *comm, hipStream_t s) {
  t_ops.push_back(Op{false, buf, count * dtype_size(dt), peer, static_cast<Comm *>(comm), s});
  if (t_depth == 0) { std::vector<Op> ops; ops.swap(t_ops); return run_group(ops); }
  return 0;
}

}  // extern "C"
