// Context helpers, model upload (reference MyModel -> device layout), synthetic generator.
#include <cstdarg>
#include <algorithm>
#include <numeric>

#include "ctx.h"

namespace lrn {

int set_error(lrn_ctx* c, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  (void)hipGetLastError();
  return code;
}

int ensure(lrn_ctx* c, DBuf& b, size_t bytes, bool zero) {
  if (bytes == 0) bytes = 8;
  if (b.bytes < bytes) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess)
      return set_error(c, LRN_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    b.bytes = bytes;
    zero = true;
  }
  if (zero) LRN_HIP(c, hipMemsetAsync(b.p, 0, b.bytes, c->stream));
  return LRN_OK;
}

void release(DBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Column-block width of the Schur sharding.  Auto: two blocks per rank (the snake map pairs a long early
// block with a short late one, so two are enough to balance the triangle), rounded up to the 128-column
// tile: the owner's GEMM3 then streams the constraint data twice instead of once per 128 columns.
void update_shard_bs(lrn_ctx* c) {
  int bs = 128;
  if (c->shard_bs_opt > 0) bs = c->shard_bs_opt;
  else if (c->world > 1 && c->nvar > 0) {
    int per = (c->nvar + 2 * c->world - 1) / (2 * c->world);
    bs = std::max(128, ((per + 127) / 128) * 128);
  }
  c->shard_bs = bs;
  c->counts["shard_bs"] = bs;
}

int copy_in(lrn_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return LRN_OK;
  if (!src || !dst) return set_error(c, LRN_ERR_ARG, "copy_in: null pointer");
  hipMemcpyKind k = is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  LRN_HIP(c, hipMemcpyAsync(dst, src, bytes, k, c->stream));
  if (k == hipMemcpyHostToDevice) LRN_HIP(c, hipStreamSynchronize(c->stream));
  return LRN_OK;
}

int copy_out(lrn_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return LRN_OK;
  if (!src || !dst) return set_error(c, LRN_ERR_ARG, "copy_out: null pointer");
  hipMemcpyKind k = is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  LRN_HIP(c, hipMemcpyAsync(dst, src, bytes, k, c->stream));
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  return LRN_OK;
}

void tic(lrn_ctx* c) {
  if (c->profile) (void)hipEventRecord(c->ev0, c->stream);
}
void toc(lrn_ctx* c, const char* key) {
  if (!c->profile) return;
  (void)hipEventRecord(c->ev1, c->stream);
  (void)hipEventSynchronize(c->ev1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  c->timing[key] += ms;
  c->counts[key] += 1;
}

// ------------------------------------------------------------------ kernels
__global__ void scatter_dense_kernel(const long* __restrict__ ptr, const int* __restrict__ er,
                                     const int* __restrict__ ec, const double* __restrict__ ev,
                                     double* __restrict__ Ad, int msz) {
  int s = blockIdx.y;   // dense slot = position
  long b = ptr[s], e = ptr[s + 1];
  double* A = Ad + (long)s * msz * msz;
  for (long k = b + (long)blockIdx.x * blockDim.x + threadIdx.x; k < e; k += (long)gridDim.x * blockDim.x)
    A[(long)er[k] + (long)ec[k] * msz] = ev[k];
}

// Philox4x32-10 (Salmon et al. 2011), counter = (k, hi, lo, 0), key = seed
__device__ inline void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ inline double synth_entry(uint64_t seed, int k, int r, int cc) {
  int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
  uint32_t c[4] = {(uint32_t)k, (uint32_t)hi, (uint32_t)lo, 0u};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double two32 = 4294967296.0;
  double u1 = ((double)c[0] + 0.5) / two32, u2 = ((double)c[1] + 0.5) / two32;
  double u3 = ((double)c[2] + 0.5) / two32, u4 = ((double)c[3] + 0.5) / two32;
  double z0 = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
  double z1 = sqrt(-2.0 * log(u3)) * cospi(2.0 * u4);
  return hi == lo ? z0 : 0.5 * (z0 + z1);
}

__global__ void synth_dense_kernel(double* __restrict__ Ad, int msz, int k0, int nk, uint64_t seed) {
  long per = (long)msz * msz;
  long total = per * nk;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int kk = (int)(e / per);
    long q = e - (long)kk * per;
    int r = (int)(q % msz), cc = (int)(q / msz);
    Ad[(long)(k0 + kk) * per + q] = synth_entry(seed, k0 + kk, r, cc);
  }
}

__global__ void build_constraint_kernel(const long* __restrict__ ptr, const int* __restrict__ er,
                                        const int* __restrict__ ec, const double* __restrict__ ev,
                                        int pos, double* __restrict__ out, int msz) {
  long b = ptr[pos], e = ptr[pos + 1];
  for (long k = b + (long)blockIdx.x * blockDim.x + threadIdx.x; k < e; k += (long)gridDim.x * blockDim.x)
    out[(long)er[k] + (long)ec[k] * msz] = ev[k];
}

}  // namespace lrn

using namespace lrn;

static void free_block(LmiBlock& b) {
  release(b.tri_tab);
  for (DBuf* d : {&b.ent_ptr, &b.ent_r, &b.ent_c, &b.ent_v, &b.Adense, &b.hidx, &b.sigma_d, &b.ipos_d, &b.cq_q, &b.cq_ptr, &b.cq_j, &b.cq_v, &b.pc_ptr, &b.pc_r, &b.pc_t, &b.ent_t, &b.Mv, &b.Zs, &b.b_ptr, &b.b_col,
                  &b.b_val, &b.X, &b.S, &b.W, &b.G, &b.Gi, &b.Si, &b.D, &b.DDsi, &b.Vprev, &b.Cd, &b.Rd, &b.delX, &b.delS, &b.Xn, &b.Sn, &b.RNT,
                  &b.t0, &b.t1, &b.t2, &b.LXf, &b.LXt, &b.LSf, &b.Yh, &b.Zh, &b.Ki, &b.Bs, &b.TX, &b.Qm, &b.lyap, &b.Bd})
    release(*d);
}

void lrn_free_model(lrn_ctx* c) {
  for (auto& b : c->lmi) free_block(b);
  c->lmi.clear();
  for (DBuf* d : {&c->cl_ptr, &c->cl_row, &c->cl_val, &c->lin_xs, &c->H, &c->L, &c->Linv, &c->cholwork,
                  &c->v0, &c->v1, &c->v2, &c->v3, &c->P, &c->P2, &c->T, &c->slabs, &c->Hd, &c->BG, &c->m0, &c->m1, &c->m2, &c->cgbuf, &c->cl_rown,
                  &c->hdiag, &c->wchol, &c->lp_r, &c->lp_c, &c->lp_ptr, &c->lp_l, &c->lp_w, &c->cr_ptr, &c->cr_col, &c->cr_val})
    release(*d);
  c->T_m = 0;
  c->T_owner = nullptr;
  c->T_layout = 0;
  c->have_H = c->have_L = false;
  c->nlmi = c->nvar = c->nlin = 0;
}


static int alloc_common(lrn_ctx* c) {
  size_t n = (size_t)c->nvar;
  LRN_TRY(ensure(c, c->H, n * n * 8, true));
  LRN_TRY(ensure(c, c->info_dev, 64, true));
  for (DBuf* d : {&c->v0, &c->v1, &c->v2, &c->v3}) LRN_TRY(ensure(c, *d, (n + 64) * 8, true));
  for (auto& b : c->lmi) {
    size_t mm = (size_t)b.msz * b.msz * 8;
    for (DBuf* d : {&b.X, &b.S, &b.W, &b.G, &b.Gi, &b.Si}) LRN_TRY(ensure(c, *d, mm, true));
    LRN_TRY(ensure(c, b.D, (size_t)b.msz * 8, true));
    LRN_TRY(ensure(c, b.DDsi, (size_t)b.msz * 8, true));
  }
  return LRN_OK;
}

extern "C" int lrn_upload_model(lrn_ctx* c, int nlmi, int nvar, const int64_t* msizes,
                                const int64_t* const* AA_colptr, const int64_t* const* AA_rowval,
                                const double* const* AA_nzval, const int64_t* const* B_colptr,
                                const int64_t* const* B_rowval, const double* const* B_nzval,
                                const int64_t* sigmaA, const int64_t* qA, int nlin,
                                const int64_t* Clin_colptr, const int64_t* Clin_rowval,
                                const double* Clin_nzval) {
  if (!c) return LRN_ERR_ARG;
  if (nlmi < 0 || nvar <= 0 || nlin < 0) return set_error(c, LRN_ERR_ARG, "bad sizes");
  if (nlmi > 0 && (!msizes || !AA_colptr || !AA_rowval || !AA_nzval || !sigmaA || !qA))
    return set_error(c, LRN_ERR_ARG, "null model array");
  LRN_HIP(c, hipSetDevice(c->device));
  lrn_free_model(c);
  c->nlmi = nlmi;
  c->nvar = nvar;
  update_shard_bs(c);
  c->nlin = nlin;
  c->pos_space = (nlmi == 1);
  c->lmi.resize(nlmi);
  for (int il = 0; il < nlmi; ++il) {
    LmiBlock& b = c->lmi[il];
    const int m = (int)msizes[il];
    if (m <= 0) return set_error(c, LRN_ERR_ARG, "msizes[%d] = %d", il, m);
    b.msz = m;
    const long ncol = (long)m * m;
    const int64_t* cp = AA_colptr[il];
    const int64_t* rv = AA_rowval[il];
    const double* nz = AA_nzval[il];
    // per-constraint counts
    std::vector<long> cnt(nvar, 0);
    long nnz_total = cp[ncol] - cp[0];
    for (long q = 0; q < ncol; ++q)
      for (long k = cp[q] - 1; k < cp[q + 1] - 1; ++k) {
        long j = rv[k] - 1;
        if (j < 0 || j >= nvar) return set_error(c, LRN_ERR_ARG, "AA rowval out of range");
        if (nz[k] != 0.0) cnt[j]++;
      }
    (void)nnz_total;
    b.sigma.resize(nvar);
    b.ipos.assign(nvar, -1);
    b.nnz.resize(nvar);
    for (int p = 0; p < nvar; ++p) {
      long s = sigmaA[(long)il * nvar + p] - 1;
      if (s < 0 || s >= nvar || b.ipos[s] != -1) return set_error(c, LRN_ERR_ARG, "sigmaA is not a permutation");
      b.sigma[p] = (int)s;
      b.ipos[s] = p;
      b.nnz[p] = cnt[s];
    }
    for (int p = 1; p < nvar; ++p)
      if (b.nnz[p] > b.nnz[p - 1])
        return set_error(c, LRN_ERR_ARG, "sigmaA is not sorted by decreasing nnz (model.jl:159)");
    b.qA = (int)qA[2 * il];
    if (b.qA < 0) b.qA = 0;
    if (b.qA > nvar) b.qA = nvar;
    b.npos_nz = 0;
    while (b.npos_nz < nvar && b.nnz[b.npos_nz] > 0) b.npos_nz++;
    // entry lists in position order
    std::vector<long> ptr(nvar + 1, 0);
    for (int p = 0; p < nvar; ++p) ptr[p + 1] = ptr[p] + b.nnz[p];
    b.nent = ptr[nvar];
    std::vector<int> er(b.nent), ec(b.nent);
    std::vector<double> ev(b.nent);
    std::vector<long> fill(ptr.begin(), ptr.end() - 1);
    for (long q = 0; q < ncol; ++q)
      for (long k = cp[q] - 1; k < cp[q + 1] - 1; ++k) {
        if (nz[k] == 0.0) continue;
        int p = b.ipos[rv[k] - 1];
        long w = fill[p]++;
        er[w] = (int)(q % m);
        ec[w] = (int)(q / m);
        ev[w] = -nz[k];     // AA = -A  (model.jl:219)
      }
    // dense (MFMA) prefix by cost model: pair path cost nnz_p * suffix_nnz vs 3 m^3 GEMM work
    {
      std::vector<double> suffix(nvar + 1, 0.0);
      for (int p = nvar - 1; p >= 0; --p) suffix[p] = suffix[p + 1] + (double)b.nnz[p];
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      long max_dense = (long)((double)free_b * 0.55 / ((double)m * m * 8.0));
      int nd = 0;
      for (int p = 0; p < b.qA && p < b.npos_nz; ++p) {
        bool dense;
        if (c->opt.dense_threshold >= 0) {
          dense = (double)b.nnz[p] >= c->opt.dense_threshold;
        } else {
          double pair_cost = (double)b.nnz[p] * suffix[p] / 1.0e11;
          double dense_cost = 3.0 * m * (double)m * m / 2.0e13 + (double)(nvar - p) * m * (double)m / 2.0e13 + 2e-5;
          dense = pair_cost > dense_cost;
        }
        if (!dense || nd >= max_dense) break;
        nd = p + 1;
      }
      b.nd = nd;
    }
    b.q_wave = b.nd;
    while (b.q_wave < b.npos_nz && b.nnz[b.q_wave] > 4) b.q_wave++;
    {  // stored columns of AA restricted to the non-dense constraints
      std::vector<long> cq_q, cq_ptr(1, 0);
      std::vector<int> cq_j;
      std::vector<double> cq_v;
      for (long q = 0; q < ncol; ++q) {
        long before = (long)cq_j.size();
        for (long k = cp[q] - 1; k < cp[q + 1] - 1; ++k) {
          if (nz[k] == 0.0) continue;
          long j = rv[k] - 1;
          if (b.ipos[j] < b.nd) continue;
          cq_j.push_back((int)j);
          cq_v.push_back(nz[k]);
        }
        if ((long)cq_j.size() > before) { cq_q.push_back(q); cq_ptr.push_back((long)cq_j.size()); }
      }
      b.ncq = (long)cq_q.size();
      LRN_TRY(ensure(c, b.cq_q, (size_t)b.ncq * 8));
      LRN_TRY(ensure(c, b.cq_ptr, (size_t)(b.ncq + 1) * 8));
      LRN_TRY(ensure(c, b.cq_j, cq_j.size() * 4));
      LRN_TRY(ensure(c, b.cq_v, cq_v.size() * 8));
      LRN_TRY(copy_in(c, b.cq_q.p, cq_q.data(), (size_t)b.ncq * 8));
      LRN_TRY(copy_in(c, b.cq_ptr.p, cq_ptr.data(), (size_t)(b.ncq + 1) * 8));
      LRN_TRY(copy_in(c, b.cq_j.p, cq_j.data(), cq_j.size() * 4));
      LRN_TRY(copy_in(c, b.cq_v.p, cq_v.data(), cq_v.size() * 8));
      // pattern of mat(AA'x) for the sparse-aware mat-vec
      b.sp_ok = false;
      if (b.nd == 0 && b.ncq > 0 && b.ncq < 2000000000L && b.nent < 2000000000L) {
        const long nq = b.ncq;
        std::vector<long> pcp(m + 1, 0);
        std::vector<int> pr(nq), pt(nq), et(b.nent);
        for (long t = 0; t < nq; ++t) {
          pr[t] = (int)(cq_q[t] % m);
          pcp[cq_q[t] / m + 1]++;
        }
        for (long q = 0; q < m; ++q) pcp[q + 1] += pcp[q];
        bool sym = true;
        for (long t = 0; t < nq && sym; ++t) {
          long key = cq_q[t] / m + (cq_q[t] % m) * m;       // (col, row) swapped
          auto it = std::lower_bound(cq_q.begin(), cq_q.end(), key);
          if (it == cq_q.end() || *it != key) sym = false;
          else pt[t] = (int)(it - cq_q.begin());
        }
        for (long e = 0; e < b.nent && sym; ++e) {
          long key = (long)er[e] + (long)ec[e] * m;
          auto it = std::lower_bound(cq_q.begin(), cq_q.end(), key);
          if (it == cq_q.end() || *it != key) sym = false;
          else et[e] = (int)(it - cq_q.begin());
        }
        if (sym) {
          LRN_TRY(ensure(c, b.pc_ptr, (size_t)(m + 1) * 8));
          LRN_TRY(ensure(c, b.pc_r, (size_t)nq * 4));
          LRN_TRY(ensure(c, b.pc_t, (size_t)nq * 4));
          LRN_TRY(ensure(c, b.ent_t, (size_t)b.nent * 4));
          LRN_TRY(ensure(c, b.Mv, (size_t)nq * 8));
          LRN_TRY(ensure(c, b.Zs, (size_t)nq * 8));
          LRN_TRY(copy_in(c, b.pc_ptr.p, pcp.data(), (size_t)(m + 1) * 8));
          LRN_TRY(copy_in(c, b.pc_r.p, pr.data(), (size_t)nq * 4));
          LRN_TRY(copy_in(c, b.pc_t.p, pt.data(), (size_t)nq * 4));
          LRN_TRY(copy_in(c, b.ent_t.p, et.data(), (size_t)b.nent * 4));
          b.sp_ok = true;
          // columns of the pattern with many entries (sp_wm_long_kernel, cgops.hip)
          b.sp_long_cols.clear();
          for (long q = 0; q < m; ++q)
            if (pcp[q + 1] - pcp[q] > 64) b.sp_long_cols.push_back((int)q);
        }
      }
    }
    LRN_TRY(ensure(c, b.ent_ptr, (size_t)(nvar + 1) * 8));
    LRN_TRY(ensure(c, b.ent_r, (size_t)b.nent * 4));
    LRN_TRY(ensure(c, b.ent_c, (size_t)b.nent * 4));
    LRN_TRY(ensure(c, b.ent_v, (size_t)b.nent * 8));
    LRN_TRY(copy_in(c, b.ent_ptr.p, ptr.data(), (size_t)(nvar + 1) * 8));
    LRN_TRY(copy_in(c, b.ent_r.p, er.data(), (size_t)b.nent * 4));
    LRN_TRY(copy_in(c, b.ent_c.p, ec.data(), (size_t)b.nent * 4));
    LRN_TRY(copy_in(c, b.ent_v.p, ev.data(), (size_t)b.nent * 8));
    std::vector<int> hidx(nvar);
    for (int p = 0; p < nvar; ++p) hidx[p] = c->pos_space ? p : b.sigma[p];
    LRN_TRY(ensure(c, b.hidx, (size_t)nvar * 4));
    LRN_TRY(copy_in(c, b.hidx.p, hidx.data(), (size_t)nvar * 4));
    LRN_TRY(ensure(c, b.sigma_d, (size_t)nvar * 4));
    LRN_TRY(ensure(c, b.ipos_d, (size_t)nvar * 4));
    LRN_TRY(copy_in(c, b.sigma_d.p, b.sigma.data(), (size_t)nvar * 4));
    LRN_TRY(copy_in(c, b.ipos_d.p, b.ipos.data(), (size_t)nvar * 4));
    if (b.nd > 0) {
      LRN_TRY(ensure(c, b.Adense, (size_t)b.nd * m * m * 8, true));
      hipLaunchKernelGGL(scatter_dense_kernel, dim3(64, b.nd), dim3(256), 0, c->stream,
                         b.ent_ptr.as<long>(), b.ent_r.as<int>(), b.ent_c.as<int>(), b.ent_v.as<double>(),
                         b.Adense.as<double>(), m);
    }
    // rank-one factors, rows reordered to H index order
    b.has_B = false;
    if (B_colptr && B_colptr[il] && B_rowval && B_nzval) {
      const int64_t* bc = B_colptr[il];
      const int64_t* br = B_rowval[il];
      const double* bv = B_nzval[il];
      std::vector<long> bcnt(nvar, 0);
      for (long q = 0; q < m; ++q)
        for (long k = bc[q] - 1; k < bc[q + 1] - 1; ++k) bcnt[br[k] - 1]++;
      // row index in H space
      std::vector<long> bptr(nvar + 1, 0);
      std::vector<int> hrow(nvar);
      for (int j = 0; j < nvar; ++j) hrow[j] = c->pos_space ? b.ipos[j] : j;
      std::vector<long> cnt_h(nvar, 0);
      for (int j = 0; j < nvar; ++j) cnt_h[hrow[j]] = bcnt[j];
      for (int h = 0; h < nvar; ++h) bptr[h + 1] = bptr[h] + cnt_h[h];
      b.bnnz = bptr[nvar];
      std::vector<int> bcol(b.bnnz);
      std::vector<double> bval(b.bnnz);
      std::vector<long> bf(bptr.begin(), bptr.end() - 1);
      for (long q = 0; q < m; ++q)
        for (long k = bc[q] - 1; k < bc[q + 1] - 1; ++k) {
          int h = hrow[br[k] - 1];
          long w = bf[h]++;
          bcol[w] = (int)q;
          bval[w] = bv[k];
        }
      LRN_TRY(ensure(c, b.b_ptr, (size_t)(nvar + 1) * 8));
      LRN_TRY(ensure(c, b.b_col, (size_t)b.bnnz * 4));
      LRN_TRY(ensure(c, b.b_val, (size_t)b.bnnz * 8));
      LRN_TRY(copy_in(c, b.b_ptr.p, bptr.data(), (size_t)(nvar + 1) * 8));
      LRN_TRY(copy_in(c, b.b_col.p, bcol.data(), (size_t)b.bnnz * 4));
      LRN_TRY(copy_in(c, b.b_val.p, bval.data(), (size_t)b.bnnz * 8));
      b.has_B = b.bnnz > 0;
    }
  }
  if (nlin > 0) {
    if (!Clin_colptr || !Clin_rowval || !Clin_nzval) return set_error(c, LRN_ERR_ARG, "null C_lin");
    long nn = Clin_colptr[nlin] - 1;
    std::vector<long> ptr(nlin + 1);
    for (int l = 0; l <= nlin; ++l) ptr[l] = Clin_colptr[l] - 1;
    std::vector<int> row(nn), rown(nn);
    for (long k = 0; k < nn; ++k) {
      long i = Clin_rowval[k] - 1;
      if (i < 0 || i >= nvar) return set_error(c, LRN_ERR_ARG, "C_lin rowval out of range");
      row[k] = c->pos_space ? c->lmi[0].ipos[i] : (int)i;
      rown[k] = (int)i;
    }
    LRN_TRY(ensure(c, c->cl_rown, (size_t)nn * 4));
    LRN_TRY(copy_in(c, c->cl_rown.p, rown.data(), (size_t)nn * 4));
    LRN_TRY(ensure(c, c->cl_ptr, (size_t)(nlin + 1) * 8));
    LRN_TRY(ensure(c, c->cl_row, (size_t)nn * 4));
    LRN_TRY(ensure(c, c->cl_val, (size_t)nn * 8));
    LRN_TRY(copy_in(c, c->cl_ptr.p, ptr.data(), (size_t)(nlin + 1) * 8));
    LRN_TRY(copy_in(c, c->cl_row.p, row.data(), (size_t)nn * 4));
    LRN_TRY(copy_in(c, c->cl_val.p, Clin_nzval, (size_t)nn * 8));
    LRN_TRY(ensure(c, c->lin_xs, (size_t)nlin * 8, true));
    {  // gather lists: every lower-triangle target (ri >= rj) of C_lin diag(xs) C_lin' with its (l, C_il C_jl)
      struct Con { int r, c, l; double w; };
      std::vector<Con> cons;
      for (int l = 0; l < nlin; ++l)
        for (long a = ptr[l]; a < ptr[l + 1]; ++a)
          for (long bq = ptr[l]; bq < ptr[l + 1]; ++bq)
            if (row[a] >= row[bq]) cons.push_back({row[a], row[bq], l, Clin_nzval[a] * Clin_nzval[bq]});
      std::stable_sort(cons.begin(), cons.end(), [](const Con& x, const Con& y) { return x.c != y.c ? x.c < y.c : x.r < y.r; });
      std::vector<int> pr, pc, pl(cons.size());
      std::vector<long> pp(1, 0);
      std::vector<double> pw(cons.size());
      for (size_t k = 0; k < cons.size(); ++k) {
        if (k == 0 || cons[k].r != cons[k - 1].r || cons[k].c != cons[k - 1].c) {
          if (k > 0) pp.push_back((long)k);
          pr.push_back(cons[k].r); pc.push_back(cons[k].c);
        }
        pl[k] = cons[k].l; pw[k] = cons[k].w;
      }
      pp.push_back((long)cons.size());
      c->lp_n = (long)pr.size();
      LRN_TRY(ensure(c, c->lp_r, pr.size() * 4 + 4));
      LRN_TRY(ensure(c, c->lp_c, pc.size() * 4 + 4));
      LRN_TRY(ensure(c, c->lp_ptr, pp.size() * 8));
      LRN_TRY(ensure(c, c->lp_l, pl.size() * 4 + 4));
      LRN_TRY(ensure(c, c->lp_w, pw.size() * 8 + 8));
      LRN_TRY(copy_in(c, c->lp_r.p, pr.data(), pr.size() * 4));
      LRN_TRY(copy_in(c, c->lp_c.p, pc.data(), pc.size() * 4));
      LRN_TRY(copy_in(c, c->lp_ptr.p, pp.data(), pp.size() * 8));
      LRN_TRY(copy_in(c, c->lp_l.p, pl.data(), pl.size() * 4));
      LRN_TRY(copy_in(c, c->lp_w.p, pw.data(), pw.size() * 8));
      // C_lin by natural rows
      std::vector<long> rp(nvar + 1, 0);
      for (long k = 0; k < nn; ++k) rp[rown[k] + 1]++;
      for (int i = 0; i < nvar; ++i) rp[i + 1] += rp[i];
      std::vector<long> fill(rp.begin(), rp.end() - 1);
      std::vector<int> rc(nn);
      std::vector<double> rv(nn);
      for (int l = 0; l < nlin; ++l)
        for (long k = ptr[l]; k < ptr[l + 1]; ++k) {
          long w = fill[rown[k]]++;
          rc[w] = l; rv[w] = Clin_nzval[k];
        }
      LRN_TRY(ensure(c, c->cr_ptr, (size_t)(nvar + 1) * 8));
      LRN_TRY(ensure(c, c->cr_col, (size_t)nn * 4 + 4));
      LRN_TRY(ensure(c, c->cr_val, (size_t)nn * 8 + 8));
      LRN_TRY(copy_in(c, c->cr_ptr.p, rp.data(), (size_t)(nvar + 1) * 8));
      LRN_TRY(copy_in(c, c->cr_col.p, rc.data(), (size_t)nn * 4));
      LRN_TRY(copy_in(c, c->cr_val.p, rv.data(), (size_t)nn * 8));
    }
  }
  LRN_TRY(alloc_common(c));
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  return LRN_OK;
}

extern "C" int lrn_synthetic_dense_model(lrn_ctx* c, int msz, int nvar, uint64_t seed) {
  if (!c || msz <= 0 || nvar <= 0) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  lrn_free_model(c);
  c->nlmi = 1;
  c->nvar = nvar;
  update_shard_bs(c);
  c->nlin = 0;
  c->pos_space = true;
  c->lmi.resize(1);
  LmiBlock& b = c->lmi[0];
  b.msz = msz;
  b.sigma.resize(nvar);
  std::iota(b.sigma.begin(), b.sigma.end(), 0);
  b.ipos = b.sigma;
  b.nnz.assign(nvar, (long)msz * msz);
  b.qA = b.nd = b.q_wave = b.npos_nz = nvar;
  b.sp_ok = false;
  b.dense_sym = 1;          // (R_k + R_k')/2 from one Philox draw per unordered index pair: symmetric by construction
  b.nent = 0;
  std::vector<long> ptr(nvar + 1, 0);
  LRN_TRY(ensure(c, b.ent_ptr, (size_t)(nvar + 1) * 8));
  LRN_TRY(copy_in(c, b.ent_ptr.p, ptr.data(), (size_t)(nvar + 1) * 8));
  LRN_TRY(ensure(c, b.ent_r, 8));
  LRN_TRY(ensure(c, b.ent_c, 8));
  LRN_TRY(ensure(c, b.ent_v, 8));
  LRN_TRY(ensure(c, b.hidx, (size_t)nvar * 4));
  LRN_TRY(copy_in(c, b.hidx.p, b.sigma.data(), (size_t)nvar * 4));
  LRN_TRY(ensure(c, b.sigma_d, (size_t)nvar * 4));
  LRN_TRY(ensure(c, b.ipos_d, (size_t)nvar * 4));
  LRN_TRY(copy_in(c, b.sigma_d.p, b.sigma.data(), (size_t)nvar * 4));
  LRN_TRY(copy_in(c, b.ipos_d.p, b.ipos.data(), (size_t)nvar * 4));
  LRN_TRY(ensure(c, b.Adense, (size_t)nvar * msz * msz * 8));
  const int chunk = 64;
  for (int k0 = 0; k0 < nvar; k0 += chunk) {
    int nk = std::min(chunk, nvar - k0);
    hipLaunchKernelGGL(synth_dense_kernel, dim3(4096), dim3(256), 0, c->stream, b.Adense.as<double>(), msz,
                       k0, nk, seed);
  }
  LRN_TRY(alloc_common(c));
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  return LRN_OK;
}

extern "C" int lrn_get_constraint(lrn_ctx* c, int ilmi, int k, double* A_out) {
  if (!c || ilmi < 0 || ilmi >= c->nlmi || k < 0 || k >= c->nvar || !A_out) return LRN_ERR_ARG;
  LmiBlock& b = c->lmi[ilmi];
  size_t mm = (size_t)b.msz * b.msz * 8;
  int pos = b.ipos[k];
  if (pos < b.nd) return copy_out(c, A_out, b.Adense.as<double>() + (size_t)pos * b.msz * b.msz, mm);
  LRN_TRY(ensure(c, c->scratch, mm));
  LRN_HIP(c, hipMemsetAsync(c->scratch.p, 0, mm, c->stream));
  hipLaunchKernelGGL(build_constraint_kernel, dim3(8), dim3(256), 0, c->stream, b.ent_ptr.as<long>(),
                     b.ent_r.as<int>(), b.ent_c.as<int>(), b.ent_v.as<double>(), pos,
                     c->scratch.as<double>(), b.msz);
  return copy_out(c, A_out, c->scratch.p, mm);
}
