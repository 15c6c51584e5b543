/* TEST DOUBLE of the ten nccl* entry points csrc/rgx_sharded.hip uses (it dlopens them: RGX_SHARDED_CCL_LIB=<this .so>), so that the
 * library's multi-RANK path -- rgx_sharded_create_rank(world > 1), the 32-byte all-gather of a round, the grouped send / recv gather,
 * a failing rank, stop requests -- runs as two, four or eight PROCESSES on a box with one GPU, where RCCL itself cannot form such a world.
 * Not a collective library: every transfer is staged through a POSIX shared-memory segment (device -> host -> shm -> host ->
 * device), ranks meet at sense-reversing barriers, send / recv pairs hand 256 KiB chunks through a mailbox per (source, destination).
 * Every wait is bounded (30 s): a protocol error in the library under test shows as ncclSystemError, never as a hang.
 *
 * Built by tests/test_gpu_sharded_capi.py:  gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/ccl_shim.c
 *                                               -L/opt/rocm/lib -lamdhip64 -lrt -o tests/_build/libccl_shim.so                      */
#define _GNU_SOURCE
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define MAXR 8
#define SLOT 4096
#define CHUNK (1 << 18)

typedef struct mbox {
  atomic_int full;
  size_t bytes;
  unsigned char data[CHUNK];
} mbox;
typedef struct shared {
  atomic_int ready, bar_count, bar_gen;
  unsigned char slot[MAXR][SLOT];
  mbox mb[MAXR][MAXR]; /* [src][dst] */
} shared;
struct ncclComm {
  shared* sh;
  int rank, n;
  char name[64];
};

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static void nap(void) {
  struct timespec t = {0, 20000};
  nanosleep(&t, NULL);
}
static void shm_name(const ncclUniqueId* id, char* out) {
  const unsigned char* b = (const unsigned char*)id->internal;
  snprintf(out, 64, "/rgx_ccl_%02x%02x%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]);
}
static int barrier(struct ncclComm* c) {
  shared* s = c->sh;
  const int gen = atomic_load(&s->bar_gen);
  if (atomic_fetch_add(&s->bar_count, 1) + 1 == c->n) {
    atomic_store(&s->bar_count, 0);
    atomic_fetch_add(&s->bar_gen, 1);
    return 0;
  }
  const double t0 = now_s();
  while (atomic_load(&s->bar_gen) == gen) {
    if (now_s() - t0 > 30.0) return -1;
    nap();
  }
  return 0;
}
static size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f || fread(id->internal, 1, 16, f) != 16) { if (f) fclose(f); return ncclSystemError; }
  fclose(f);
  char name[64];
  shm_name(id, name);
  const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)sizeof(shared)) != 0) { close(fd); return ncclSystemError; }
  close(fd);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof *c);
  shm_name(&id, c->name);
  int fd = -1;
  const double t0 = now_s();
  while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0) {
    if (now_s() - t0 > 30.0) { free(c); return ncclSystemError; }
    nap();
  }
  c->sh = (shared*)mmap(NULL, sizeof(shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->sh == MAP_FAILED) { free(c); return ncclSystemError; }
  c->rank = rank; c->n = nranks;
  atomic_fetch_add(&c->sh->ready, 1);
  while (atomic_load(&c->sh->ready) < nranks) {
    if (now_s() - t0 > 30.0) { free(c); return ncclSystemError; }
    nap();
  }
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  (void)comms; (void)ndev; (void)devlist;
  return ncclInvalidUsage; /* one process, several devices: not what this double is for */
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  struct ncclComm* c = comm;
  if (!c) return ncclSuccess;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(shared));
  free(c);
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  struct ncclComm* c = comm;
  const size_t bytes = sendcount * dtype_size(datatype);
  if (bytes > SLOT) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->sh->slot[c->rank], sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (barrier(c)) return ncclSystemError;
  for (int r = 0; r < c->n; r++)
    if (hipMemcpy((char*)recvbuff + (size_t)r * bytes, c->sh->slot[r], bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  if (barrier(c)) return ncclSystemError;
  return ncclSuccess;
}

/* ---- send / recv: queued between GroupStart and GroupEnd, then driven together (two ranks that send to each other through bounded
 * mailboxes must not wait for each other's receive) */
typedef struct op { int is_send, peer; char* dev; size_t bytes, done; struct ncclComm* c; hipStream_t st; } op;
static __thread op g_ops[64];
static __thread int g_nops = 0, g_depth = 0;

static int progress(op* o) { /* 1: moved something */
  if (o->done >= o->bytes) return 0;
  const size_t k = o->bytes - o->done < CHUNK ? o->bytes - o->done : CHUNK;
  if (o->is_send) {
    mbox* m = &o->c->sh->mb[o->c->rank][o->peer];
    if (atomic_load(&m->full)) return 0;
    if (hipMemcpy(m->data, o->dev + o->done, k, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    m->bytes = k;
    atomic_store(&m->full, 1);
  } else {
    mbox* m = &o->c->sh->mb[o->peer][o->c->rank];
    if (!atomic_load(&m->full)) return 0;
    if (m->bytes != k) return -1; /* the two sides disagree about the size: a protocol error of the library under test */
    if (hipMemcpy(o->dev + o->done, m->data, k, hipMemcpyHostToDevice) != hipSuccess) return -1;
    atomic_store(&m->full, 0);
  }
  o->done += k;
  return 1;
}
static ncclResult_t drive(void) {
  const double t0 = now_s();
  for (;;) {
    int left = 0, moved = 0;
    for (int i = 0; i < g_nops; i++) {
      const int p = progress(&g_ops[i]);
      if (p < 0) { g_nops = 0; return ncclSystemError; }
      moved |= p;
      left += g_ops[i].done < g_ops[i].bytes;
    }
    if (!left) break;
    if (!moved) {
      if (now_s() - t0 > 30.0) { g_nops = 0; return ncclSystemError; }
      nap();
    }
  }
  g_nops = 0;
  return ncclSuccess;
}
static ncclResult_t enqueue(int is_send, void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  if (g_nops >= 64 || peer < 0 || peer >= comm->n) return ncclInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError; /* the rows were produced on this stream */
  op* o = &g_ops[g_nops++];
  o->is_send = is_send; o->peer = peer; o->dev = (char*)buf; o->bytes = count * dtype_size(dt); o->done = 0; o->c = comm; o->st = st;
  return g_depth ? ncclSuccess : drive();
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(1, (void*)sendbuff, count, datatype, peer, comm, stream);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(0, recvbuff, count, datatype, peer, comm, stream);
}
ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
  if (g_depth > 0) g_depth--;
  return g_depth ? ncclSuccess : drive();
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclSystemError: return "ccl_shim: timeout or shared-memory failure";
    case ncclInvalidArgument: return "ccl_shim: invalid argument";
    case ncclInvalidUsage: return "ccl_shim: not implemented";
    default: return "ccl_shim: HIP error";
  }
}
