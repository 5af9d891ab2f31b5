/* TEST INFRASTRUCTURE -- a stand-in for librccl with the eight entry points libsapca binds by name (csrc/comm.cpp), so
 * that the library's RCCL mode (Comm::RCCL: the duplicate communicator for the side stream, the abort path, a split
 * ncclCommInitRank outcome) runs with 2-4 ranks on the ONE GPU of the test box, where the real RCCL refuses ranks that
 * share a device.  Never part of the product: built by tests/test_gpu_fake_rccl.py into a temporary directory and reached
 * only through SAPCA_RCCL_LIBRARY, which only the -DSAPCA_DEBUG_SWITCHES build of libsapca reads.
 *
 * Transport: one POSIX shared-memory segment per ncclUniqueId.  An all-reduce is synchronous on the host: wait for the
 * stream, copy the chunk to this rank's slot, barrier, every rank sums the slots in rank order (bitwise identical results
 * on all ranks), copy back, barrier.  Two communicator lanes (the one from ncclCommInitRank, one from ncclCommSplit) have
 * their own barriers and slots.  Every wait polls the communicator's abort flag and gives up after FAKE_RCCL_TIMEOUT_S.
 *
 * Hooks (environment): FAKE_RCCL_FAIL_INIT_RANK=r  -- rank r's ncclCommInitRank fails after everyone has joined (a split
 * outcome); FAKE_RCCL_NO_SPLIT_RANK=r -- rank r's ncclCommSplit fails (the ranks must still agree on the side lane);
 * FAKE_RCCL_FAILFAST=1 -- a wait gives up with ncclRemoteError as soon as a peer is gone (otherwise it spins like a
 * collective kernel would, until this rank's communicator is aborted or the timeout).
 *
 *   hipcc -O2 -shared -fPIC -x hip -o libfake_rccl.so tests/fake_rccl/fake_rccl.c
 */
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define MAX_RANKS 8
#define LANES 2
#define CHUNK_BYTES (4u << 20)
#define NCCL_SUCCESS 0
#define NCCL_SYSTEM_ERROR 2
#define NCCL_INTERNAL_ERROR 3
#define NCCL_INVALID_ARGUMENT 4
#define NCCL_REMOTE_ERROR 6

typedef struct {
  _Atomic int joined;               /* ranks inside ncclCommInitRank */
  _Atomic int gone[MAX_RANKS];      /* a rank aborted or destroyed its communicator */
  _Atomic int bar_count[LANES];     /* sense-reversing barrier per lane */
  _Atomic int bar_sense[LANES];
  _Atomic int split_arrived;
  char pad[64];
  /* then LANES x MAX_RANKS slots of CHUNK_BYTES */
} shm_head;

typedef struct fake_comm {
  shm_head* shm;
  size_t shm_bytes;
  int rank, nranks, lane;
  int local_sense;
  _Atomic int aborted;
  char name[64];
} fake_comm;

typedef struct { char internal[128]; } ncclUniqueId;

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static double timeout_s(void) {
  const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 60.0;
}
static size_t seg_bytes(void) { return sizeof(shm_head) + (size_t)LANES * MAX_RANKS * CHUNK_BYTES; }
static char* slot(fake_comm* c, int rank) { return (char*)c->shm + sizeof(shm_head) + ((size_t)c->lane * MAX_RANKS + rank) * CHUNK_BYTES; }

/* 0: passed; otherwise an ncclResult (aborted locally, a peer is gone, or the timeout) */
static int barrier(fake_comm* c) {
  shm_head* h = c->shm;
  const int sense = c->local_sense ^= 1;
  if (atomic_fetch_add(&h->bar_count[c->lane], 1) == c->nranks - 1) {
    atomic_store(&h->bar_count[c->lane], 0);
    atomic_store(&h->bar_sense[c->lane], sense);
    return NCCL_SUCCESS;
  }
  const double t0 = now_s();
  const int failfast = getenv("FAKE_RCCL_FAILFAST") != NULL;
  while (atomic_load(&h->bar_sense[c->lane]) != sense) {
    if (atomic_load(&c->aborted)) return NCCL_INTERNAL_ERROR;
    if (failfast)
      for (int r = 0; r < c->nranks; ++r)
        if (atomic_load(&h->gone[r])) return NCCL_REMOTE_ERROR;
    if (now_s() - t0 > timeout_s()) return NCCL_SYSTEM_ERROR;
    usleep(50);
  }
  return NCCL_SUCCESS;
}

extern "C" __attribute__((visibility("default"))) int ncclGetUniqueId(ncclUniqueId* id) {
  static _Atomic int counter;
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/sapca_fake_rccl_%d_%d", (int)getpid(), atomic_fetch_add(&counter, 1));
  int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return NCCL_SYSTEM_ERROR;
  if (ftruncate(fd, (off_t)seg_bytes()) != 0) { close(fd); return NCCL_SYSTEM_ERROR; }
  close(fd);   /* (fresh pages are zero: every counter starts at 0) */
  return NCCL_SUCCESS;
}

extern "C" __attribute__((visibility("default"))) int ncclCommInitRank(fake_comm** out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return NCCL_INVALID_ARGUMENT;
  int fd = shm_open(id.internal, O_RDWR, 0600);
  if (fd < 0) return NCCL_SYSTEM_ERROR;
  void* p = mmap(NULL, seg_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return NCCL_SYSTEM_ERROR;
  fake_comm* c = (fake_comm*)calloc(1, sizeof(fake_comm));
  c->shm = (shm_head*)p; c->shm_bytes = seg_bytes(); c->rank = rank; c->nranks = nranks; c->lane = 0;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  atomic_fetch_add(&c->shm->joined, 1);
  const double t0 = now_s();
  while (atomic_load(&c->shm->joined) < nranks) {
    if (now_s() - t0 > timeout_s()) { munmap(p, seg_bytes()); free(c); return NCCL_SYSTEM_ERROR; }
    usleep(50);
  }
  if (rank == 0) shm_unlink(id.internal);   /* everyone has it mapped: the name can go */
  const char* fail = getenv("FAKE_RCCL_FAIL_INIT_RANK");
  if (fail && atoi(fail) == rank) {
    atomic_store(&c->shm->gone[rank], 1);
    return NCCL_SYSTEM_ERROR;                /* (the mapping leaks: a test double) */
  }
  *out = c;
  return NCCL_SUCCESS;
}

extern "C" __attribute__((visibility("default"))) int ncclCommSplit(fake_comm* c, int color, int key, fake_comm** out, void* config) {
  (void)color; (void)key; (void)config;
  if (!c || c->lane != 0) return NCCL_INVALID_ARGUMENT;
  int rc = barrier(c);   /* collective, like the real one */
  if (rc) return rc;
  const char* no = getenv("FAKE_RCCL_NO_SPLIT_RANK");
  if (no && atoi(no) == c->rank) return NCCL_SYSTEM_ERROR;
  fake_comm* d = (fake_comm*)calloc(1, sizeof(fake_comm));
  *d = *c;
  d->lane = 1; d->local_sense = 0; atomic_store(&d->aborted, 0);
  *out = d;
  return NCCL_SUCCESS;
}

/* every rank sums every slot in rank order into a private block (the slots are still being read by the other ranks):
 * identical bits everywhere */
static void* sum_slots(fake_comm* c, size_t n, int dtype) {
  if (dtype == 7) {
    float* acc = (float*)malloc(n * sizeof(float));
    memcpy(acc, slot(c, 0), n * sizeof(float));
    for (int r = 1; r < c->nranks; ++r) {
      const float* s = (const float*)slot(c, r);
      for (size_t i = 0; i < n; ++i) acc[i] += s[i];
    }
    return acc;
  }
  double* acc = (double*)malloc(n * sizeof(double));
  memcpy(acc, slot(c, 0), n * sizeof(double));
  for (int r = 1; r < c->nranks; ++r) {
    const double* s = (const double*)slot(c, r);
    for (size_t i = 0; i < n; ++i) acc[i] += s[i];
  }
  return acc;
}

extern "C" __attribute__((visibility("default"))) int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, fake_comm* c,
                                                          hipStream_t stream) {
  if (!c || (dtype != 7 && dtype != 8) || op != 0) return NCCL_INVALID_ARGUMENT;
  if (atomic_load(&c->aborted)) return NCCL_INTERNAL_ERROR;
  const size_t esz = dtype == 7 ? 4 : 8, per = CHUNK_BYTES / esz;
  if (hipStreamSynchronize(stream) != hipSuccess) return NCCL_SYSTEM_ERROR;   /* the producer kernels of `send` */
  for (size_t off = 0; off < count; off += per) {
    const size_t n = count - off < per ? count - off : per;
    if (hipMemcpy(slot(c, c->rank), (const char*)send + off * esz, n * esz, hipMemcpyDeviceToHost) != hipSuccess) return NCCL_SYSTEM_ERROR;
    int rc = barrier(c);
    if (rc) return rc;
    void* acc = sum_slots(c, n, dtype);
    rc = barrier(c);   /* everyone has read every slot: they may be overwritten */
    if (rc) { free(acc); return rc; }
    const hipError_t e = hipMemcpy((char*)recv + off * esz, acc, n * esz, hipMemcpyHostToDevice);
    free(acc);
    if (e != hipSuccess) return NCCL_SYSTEM_ERROR;
  }
  return NCCL_SUCCESS;
}

extern "C" __attribute__((visibility("default"))) int ncclCommAbort(fake_comm* c) {
  if (!c) return NCCL_INVALID_ARGUMENT;
  atomic_store(&c->aborted, 1);              /* whoever waits in a barrier of this communicator returns */
  atomic_store(&c->shm->gone[c->rank], 1);   /* peers see an asynchronous error */
  return NCCL_SUCCESS;                       /* (nothing is freed: a thread may still be inside a call) */
}

extern "C" __attribute__((visibility("default"))) int ncclCommDestroy(fake_comm* c) {
  if (!c) return NCCL_INVALID_ARGUMENT;
  if (c->lane == 0) {
    atomic_store(&c->shm->gone[c->rank], 1);
    munmap(c->shm, c->shm_bytes);
  }
  free(c);
  return NCCL_SUCCESS;
}

extern "C" __attribute__((visibility("default"))) int ncclCommGetAsyncError(fake_comm* c, int* err) {
  if (!c || !err) return NCCL_INVALID_ARGUMENT;
  *err = NCCL_SUCCESS;
  for (int r = 0; r < c->nranks; ++r)
    if (r != c->rank && atomic_load(&c->shm->gone[r])) *err = NCCL_REMOTE_ERROR;
  return NCCL_SUCCESS;
}

extern "C" __attribute__((visibility("default"))) const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case 0: return "no error";
    case NCCL_SYSTEM_ERROR: return "fake rccl: system error (timeout, a hook, or shared memory)";
    case NCCL_INTERNAL_ERROR: return "fake rccl: communicator aborted";
    case NCCL_INVALID_ARGUMENT: return "fake rccl: invalid argument";
    case NCCL_REMOTE_ERROR: return "fake rccl: a peer is gone";
    default: return "fake rccl: error";
  }
}
