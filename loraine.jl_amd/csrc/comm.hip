// Multi-GPU exchange of the hot path behind the C ABI (round 3): one process per GPU, one RCCL communicator per
// context.  The reference is a single process (no MPI/NCCL anywhere); what is sharded here is
// what section 8e lists -- kit=0: the Schur assembly (partial sums + all-reduce on the Cholesky path, column blocks +
// all-gather otherwise) before the replicated factorisation (src/predictor_corrector.jl:24-40,53-58); kit=1: the CG
// operator (src/Solvers.jl:572-614) with one all-reduce of an nvar-vector per application (:134,235).
//
// All collectives run on the context's stream: nothing leaves the device, the host never synchronises for them.
// A second transport with the same entry points stages through the host and calls back into the embedding program
// (lrn_comm_init_host): ranks that share one GPU, or a launcher whose only fabric is gloo/MPI on the CPU.
#include <rccl/rccl.h>

#include <algorithm>

#include "../../include/loraine_hip.h"
#include "ctx.h"

namespace lrn {

struct Comm {
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  lrn_host_allreduce_fn har = nullptr;
  lrn_host_allgather_fn hag = nullptr;
  void* user = nullptr;
  std::vector<double> hsend, hrecv;
  DBuf pack;       // lower triangle of H / exchange buffers
  DBuf gathered;
  DBuf flags;      // small device buffer for status words
  long calls = 0;
  bool fail_next_ensure = false;   // test hook (option "comm_fail_ensure"): the next exchange fails its buffer allocation
};

void comm_inject_ensure_failure(lrn_ctx* c) {
  if (c->comm) c->comm->fail_next_ensure = true;
}

static int nccl_fail(lrn_ctx* c, ncclResult_t r, const char* what) {
  return set_error(c, LRN_ERR_HIP, "%s failed: %s", what, ncclGetErrorString(r));
}

// in place, on c->stream.  op: 0 sum, 1 min, 2 max
int comm_allreduce(lrn_ctx* c, double* buf, long count, int op) {
  Comm* m = c->comm;
  if (!m || count <= 0 || (m->world <= 1 && !m->nccl)) return LRN_OK;
  m->calls += 1;
  if (m->nccl) {
    const ncclRedOp_t o = op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax);
    ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, o, m->nccl, c->stream);
    if (r != ncclSuccess) return nccl_fail(c, r, "ncclAllReduce");
    return LRN_OK;
  }
  if (!m->har) return set_error(c, LRN_ERR_STATE, "communicator has no transport");
  m->hsend.resize((size_t)count);
  LRN_HIP(c, hipMemcpyAsync(m->hsend.data(), buf, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  if (m->har(m->user, m->hsend.data(), (int64_t)count, op) != 0)
    return set_error(c, LRN_ERR_STATE, "host all-reduce callback failed");
  LRN_HIP(c, hipMemcpyAsync(buf, m->hsend.data(), (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
  LRN_HIP(c, hipStreamSynchronize(c->stream));      // (hsend may be resized by the next call)
  return LRN_OK;
}

// recv[r * count .. ) = send of rank r; device buffers, on c->stream
int comm_allgather(lrn_ctx* c, const double* send, double* recv, long count) {
  Comm* m = c->comm;
  if (!m || count <= 0) return LRN_OK;
  m->calls += 1;
  if (m->world <= 1) {
    LRN_HIP(c, hipMemcpyAsync(recv, send, (size_t)count * 8, hipMemcpyDeviceToDevice, c->stream));
    return LRN_OK;
  }
  if (m->nccl) {
    ncclResult_t r = ncclAllGather(send, recv, (size_t)count, ncclDouble, m->nccl, c->stream);
    if (r != ncclSuccess) return nccl_fail(c, r, "ncclAllGather");
    return LRN_OK;
  }
  if (!m->hag) return set_error(c, LRN_ERR_STATE, "communicator has no transport");
  m->hsend.resize((size_t)count);
  m->hrecv.resize((size_t)count * m->world);
  LRN_HIP(c, hipMemcpyAsync(m->hsend.data(), send, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  if (m->hag(m->user, m->hsend.data(), m->hrecv.data(), (int64_t)count) != 0)
    return set_error(c, LRN_ERR_STATE, "host all-gather callback failed");
  LRN_HIP(c, hipMemcpyAsync(recv, m->hrecv.data(), (size_t)count * m->world * 8, hipMemcpyHostToDevice, c->stream));
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  return LRN_OK;
}

// Column blocks of an n x n column-major matrix, in place: rank r holds columns [r cb, min(n, (r + 1) cb)) and receives
// the others (the sharded n^3 products of the resident path, prepw.hip::pgemm_nt).  RCCL: one broadcast per rank inside a
// group (the last block may be narrower); host transport: packed through the equal-count all-gather callback.
int comm_allgather_cols(lrn_ctx* c, double* C, int n, int cb) {
  Comm* m = c->comm;
  if (!m || m->world <= 1) return LRN_OK;
  m->calls += 1;
  if (m->nccl) {
    ncclResult_t r = ncclGroupStart();
    for (int k = 0; k < m->world && r == ncclSuccess; ++k) {
      const long c0 = std::min<long>(n, (long)k * cb), c1 = std::min<long>(n, c0 + cb);
      if (c1 > c0) r = ncclBroadcast(C + c0 * n, C + c0 * n, (size_t)((c1 - c0) * n), ncclDouble, k, m->nccl, c->stream);
    }
    const ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess) return nccl_fail(c, r, "ncclBroadcast");
    if (r2 != ncclSuccess) return nccl_fail(c, r2, "ncclGroupEnd");
    return LRN_OK;
  }
  const long per = (long)cb * n;
  LRN_TRY(ensure(c, m->pack, (size_t)per * 8));
  LRN_TRY(ensure(c, m->gathered, (size_t)per * m->world * 8));
  const long c0 = std::min<long>(n, (long)m->rank * cb), c1 = std::min<long>(n, c0 + cb);
  LRN_HIP(c, hipMemsetAsync(m->pack.p, 0, (size_t)per * 8, c->stream));
  if (c1 > c0)
    LRN_HIP(c, hipMemcpyAsync(m->pack.p, C + c0 * n, (size_t)((c1 - c0) * n) * 8, hipMemcpyDeviceToDevice, c->stream));
  LRN_TRY(comm_allgather(c, m->pack.as<double>(), m->gathered.as<double>(), per));
  for (int k = 0; k < m->world; ++k) {
    if (k == m->rank) continue;
    const long k0 = std::min<long>(n, (long)k * cb), k1 = std::min<long>(n, k0 + cb);
    if (k1 > k0)
      LRN_HIP(c, hipMemcpyAsync(C + k0 * n, m->gathered.as<double>() + (long)k * per, (size_t)((k1 - k0) * n) * 8,
                                hipMemcpyDeviceToDevice, c->stream));
  }
  return LRN_OK;
}

// a few status words, max-reduced over the ranks (host in, host out; every rank must call it)
int comm_status_max(lrn_ctx* c, double* words, int nw) {
  Comm* m = c->comm;
  if (!m || m->world <= 1) return LRN_OK;
  LRN_TRY(ensure(c, m->flags, 64 * 8));
  if (nw > 64) return LRN_ERR_ARG;
  LRN_HIP(c, hipMemcpyAsync(m->flags.p, words, (size_t)nw * 8, hipMemcpyHostToDevice, c->stream));
  LRN_TRY(comm_allreduce(c, m->flags.as<double>(), nw, 2));
  LRN_HIP(c, hipMemcpyAsync(words, m->flags.p, (size_t)nw * 8, hipMemcpyDeviceToHost, c->stream));
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  return LRN_OK;
}

// lower triangle of the n x n column-major H <-> packed buffer (column j: rows j..n-1)
__global__ void pack_lower_kernel(const double* __restrict__ H, int n, double* __restrict__ buf) {
  const int j = blockIdx.x;
  const long off = (long)j * n - (long)j * (j - 1) / 2;
  for (int i = j + threadIdx.x; i < n; i += blockDim.x) buf[off + (i - j)] = H[(long)i + (long)j * n];
}
__global__ void unpack_lower_kernel(const double* __restrict__ buf, int n, double* __restrict__ H) {
  const int j = blockIdx.x;
  const long off = (long)j * n - (long)j * (j - 1) / 2;
  for (int i = j + threadIdx.x; i < n; i += blockDim.x) H[(long)i + (long)j * n] = buf[off + (i - j)];
}

// The exchange after an assembly on this context's communicator.  `rc_local` is what the local assembly returned:
// every rank enters the status reduction whatever happened to it, so a rank that failed (out of memory, a W that could
// not be factored where the others could) makes ALL ranks return an error instead of leaving its peers in a collective.
int comm_schur_exchange(lrn_ctx* c, int rc_local, bool gather_blocks) {
  Comm* m = c->comm;
  if (!m || m->world <= 1) return rc_local;
  const int n = c->nvar;
  // The exchange buffers are allocated BEFORE the status reduction and their failure is part of the status word: a rank
  // that cannot allocate them must not return while its peers enter the collective below.
  const long np = (long)n * (n + 1) / 2;
  const long per = (long)lrn_schur_shard_doubles(c);
  int rc_buf = LRN_OK;
  if (rc_local == LRN_OK) {
    if (m->fail_next_ensure) { m->fail_next_ensure = false; rc_buf = set_error(c, LRN_ERR_NOMEM, "exchange buffers: injected allocation failure (test)"); }
    else if (c->H_partial) rc_buf = ensure(c, m->pack, (size_t)np * 8);
    else if (gather_blocks) {
      rc_buf = ensure(c, m->pack, (size_t)per * 8);
      if (rc_buf == LRN_OK) rc_buf = ensure(c, m->gathered, (size_t)per * m->world * 8);
    }
  }
  const int rc_mine = rc_local != LRN_OK ? rc_local : rc_buf;
  const std::string err_mine = c->err;
  double w[3] = {rc_mine == LRN_OK ? 0.0 : 1.0, c->H_partial ? 1.0 : 0.0, c->H_partial ? 0.0 : 1.0};
  int rs = comm_status_max(c, w, 3);
  if (rs != LRN_OK) return rs;
  if (w[0] != 0.0) {
    if (rc_mine != LRN_OK) { c->err = err_mine; return rc_mine; }
    return set_error(c, LRN_ERR_STATE, "Schur assembly failed on another rank");
  }
  if (w[1] != 0.0 && w[2] != 0.0)
    return set_error(c, LRN_ERR_STATE, "ranks disagree on the Schur exchange (partial sums on some, column blocks on others)");
  c->H_owned_only = false;
  if (!c->H_partial && !gather_blocks) {      // the CG operator multiplies the columns this rank assembled: nothing to exchange
    c->H_owned_only = true;
    return LRN_OK;
  }
  hipEvent_t a0, a1;
  if (c->profile) { (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventRecord(a0, c->stream); }
  if (c->H_partial) {
    // partial sums of the whole matrix: only the lower triangle is authoritative -> n (n + 1) / 2 doubles
    hipLaunchKernelGGL(pack_lower_kernel, dim3(n), dim3(256), 0, c->stream, c->H.as<double>(), n, m->pack.as<double>());
    LRN_TRY(comm_allreduce(c, m->pack.as<double>(), np, 0));
    hipLaunchKernelGGL(unpack_lower_kernel, dim3(n), dim3(256), 0, c->stream, m->pack.as<double>(), n, c->H.as<double>());
    c->H_partial = false;
  } else {
    LRN_TRY(lrn_schur_export_shard(c, m->pack.as<double>()));
    LRN_TRY(comm_allgather(c, m->pack.as<double>(), m->gathered.as<double>(), per));
    LRN_TRY(lrn_schur_import_all(c, m->gathered.as<double>()));
  }
  if (c->profile) {
    (void)hipEventRecord(a1, c->stream); (void)hipEventSynchronize(a1);
    float ms = 0; (void)hipEventElapsedTime(&ms, a0, a1);
    c->timing["exchange"] += ms; c->counts["exchange"] += 1;
    (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
  }
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

// Before the first sharded assembly: the ranks agree on the assembly path (it decides which collective follows) --
// all-reduce MIN of every rank's own view, pinned as option schur_plan.
int comm_agree_plan(lrn_ctx* c, int mode) {
  Comm* m = c->comm;
  if (!m || m->world <= 1 || c->opt.schur_plan >= 0) return LRN_OK;
  double w[1] = {-(double)schur_plan(c, mode)};        // max of the negatives = -min
  LRN_TRY(comm_status_max(c, w, 1));
  c->opt.schur_plan = (int)(-w[0] + 0.5);
  c->counts["schur_plan_agreed"] = c->opt.schur_plan;
  return LRN_OK;
}

void comm_free(lrn_ctx* c) {
  Comm* m = c->comm;
  if (!m) return;
  if (m->nccl) (void)ncclCommDestroy(m->nccl);
  release(m->pack); release(m->gathered); release(m->flags);
  delete m;
  c->comm = nullptr;
}

}  // namespace lrn

using namespace lrn;

extern "C" int lrn_comm_unique_id(void* id128) {
  if (!id128) return LRN_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return LRN_ERR_HIP;
  memcpy(id128, &id, 128);
  return LRN_OK;
}

extern "C" int lrn_comm_init(lrn_ctx* c, const void* id128, int rank, int world) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  comm_free(c);
  Comm* m = new Comm();
  m->rank = rank; m->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = ncclCommInitRank(&m->nccl, world, id, rank);
  if (r != ncclSuccess) { delete m; return nccl_fail(c, r, "ncclCommInitRank"); }
  c->comm = m;
  return lrn_set_shard(c, rank, world);
}

extern "C" int lrn_comm_init_host(lrn_ctx* c, int rank, int world, lrn_host_allreduce_fn allreduce,
                                  lrn_host_allgather_fn allgather, void* user) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && (!allreduce || !allgather))) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  comm_free(c);
  Comm* m = new Comm();
  m->rank = rank; m->world = world;
  m->har = allreduce; m->hag = allgather; m->user = user;
  c->comm = m;
  return lrn_set_shard(c, rank, world);
}

extern "C" int lrn_comm_destroy(lrn_ctx* c) {
  if (!c) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  comm_free(c);
  return lrn_set_shard(c, 0, 1);
}

extern "C" int lrn_comm_allreduce(lrn_ctx* c, double* buf, int64_t count, int op) {
  if (!c || !buf || count < 0 || op < 0 || op > 2) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  if (!c->comm) return set_error(c, LRN_ERR_STATE, "no communicator (lrn_comm_init)");
  if (is_device_ptr(buf)) {
    LRN_TRY(comm_allreduce(c, buf, (long)count, op));
    LRN_HIP(c, hipStreamSynchronize(c->stream));
    return LRN_OK;
  }
  DBuf d;
  LRN_TRY(ensure(c, d, (size_t)count * 8));
  int rc = copy_in(c, d.p, buf, (size_t)count * 8);
  if (rc == LRN_OK) rc = comm_allreduce(c, d.as<double>(), (long)count, op);
  if (rc == LRN_OK) rc = copy_out(c, buf, d.p, (size_t)count * 8);
  release(d);
  return rc;
}
