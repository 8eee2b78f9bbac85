"""Thin object wrapper over the C ABI (include/loraine_hip.h): one `Device` = one `lrn_ctx`
= one GPU of one process.  Array arguments are numpy arrays (host) or torch CUDA tensors /
raw device addresses; nothing here computes -- every method is one C-ABI call."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _capi
from ._capi import LoraineHipError, f64, ptr, ptr_array


class Device:
    def __init__(self, device: int = 0):
        self.lib = _capi.load_library()
        if self.lib.lrn_device_count() <= 0:
            raise LoraineHipError("no HIP device visible: the Loraine hot path runs on MI355X only "
                                  "(there is no CPU fallback)")
        h = _capi.c_ctx()
        rc = self.lib.lrn_create(C.byref(h), int(device))
        if rc != 0:
            raise LoraineHipError(f"lrn_create(device={device}) failed with {rc}")
        self.h = h
        self.nvar = 0
        self.msizes = []
        self.nlin = 0

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.lrn_last_error(self.h)
            raise LoraineHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.lrn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._chk(self.lib.lrn_set_option(self.h, key.encode(), float(value)), f"set_option({key})")

    def shard_bs(self):
        """column-block width of the Schur sharding in effect (auto: two blocks per rank, multiple of 128)"""
        return int(self.count("shard_bs"))

    def set_shard(self, rank, world):
        self._chk(self.lib.lrn_set_shard(self.h, int(rank), int(world)), "set_shard")

    # ------------------------------------------------------------------ multi-GPU communicator (csrc/comm.hip)
    def comm_unique_id(self):
        """RCCL unique id (128 bytes) -- rank 0 creates it, the launcher hands it to every rank."""
        buf = (C.c_ubyte * 128)()
        rc = self.lib.lrn_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise LoraineHipError(f"lrn_comm_unique_id failed ({rc})")
        return bytes(buf)

    def comm_init(self, uid, rank, world):
        """ncclCommInitRank on this context's device: the exchanges of lrn_schur_assemble / lrn_pcg / lrn_matvec run
        inside the library from now on."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(uid))
        self._chk(self.lib.lrn_comm_init(self.h, C.cast(buf, C.c_void_p), int(rank), int(world)), "lrn_comm_init")

    def comm_init_host(self, rank, world, allreduce, allgather):
        """The same entry points over host callbacks: allreduce(np_array, op) reduces in place (op 0 sum, 1 min,
        2 max), allgather(send, recv) fills recv (world * len(send))."""
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int)
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)

        def _ar(_user, buf, count, op):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)), int(op))
                return 0
            except Exception as e:          # noqa: BLE001 -- an exception must not unwind through the C frames
                print(f"[loraine.jl_amd] host all-reduce callback: {e!r}", flush=True)
                return 1

        def _ag(_user, send, recv, count):
            try:
                allgather(np.ctypeslib.as_array(send, shape=(int(count),)),
                          np.ctypeslib.as_array(recv, shape=(int(count) * int(world),)))
                return 0
            except Exception as e:          # noqa: BLE001
                print(f"[loraine.jl_amd] host all-gather callback: {e!r}", flush=True)
                return 1

        self._comm_cb = (AR(_ar), AG(_ag))              # keep the trampolines alive as long as the communicator
        self._chk(self.lib.lrn_comm_init_host(self.h, int(rank), int(world), C.cast(self._comm_cb[0], C.c_void_p),
                                              C.cast(self._comm_cb[1], C.c_void_p), None), "lrn_comm_init_host")

    def comm_init_torch(self, rank, world, group=None):
        """Create the library's communicator from a torch.distributed process group: RCCL when the group's backend is
        nccl (the unique id travels through the group), host-staged callbacks over the group otherwise (gloo: ranks
        that share a GPU, CPU-only fabrics)."""
        import torch
        import torch.distributed as dist
        if world <= 1 and not dist.is_initialized():
            self.comm_init(self.comm_unique_id(), 0, 1)
            return "rccl"
        if dist.get_backend(group) == "nccl":
            box = [self.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            self.comm_init(box[0], rank, world)
            return "rccl"
        ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}

        def ar(a, op):
            dist.all_reduce(torch.from_numpy(a), op=ops[op], group=group)

        def ag(send, recv):
            dist.all_gather_into_tensor(torch.from_numpy(recv), torch.from_numpy(send), group=group)

        self.comm_init_host(rank, world, ar, ag)
        return "host"

    def comm_destroy(self):
        self._chk(self.lib.lrn_comm_destroy(self.h), "lrn_comm_destroy")
        self._comm_cb = None

    def comm_allreduce(self, arr, op=0):
        arr = f64(arr)
        self._chk(self.lib.lrn_comm_allreduce(self.h, ptr(arr), int(arr.size), int(op)), "lrn_comm_allreduce")
        return arr

    def timing(self, key):
        v = C.c_double(0.0)
        self.lib.lrn_get_timing(self.h, key.encode(), C.byref(v))
        return v.value

    def count(self, key):
        return int(self.lib.lrn_get_count(self.h, key.encode()))

    def reset_timing(self):
        self.set_option("reset_timing", 1)

    # ------------------------------------------------------------------ model
    @staticmethod
    def _csc64(M):
        """scipy sparse -> (colptr, rowval, nzval), Int64, 1-based (Julia SparseMatrixCSC)."""
        M = sp.csc_matrix(M)
        M.sort_indices()
        return ((M.indptr.astype(np.int64) + 1), (M.indices.astype(np.int64) + 1),
                np.ascontiguousarray(M.data, dtype=np.float64))

    def upload_model(self, AA, sigmaA, qA, msizes, B=None, C_lin=None):
        """AA: list of (nvar x msz^2) sparse; sigmaA (nvar x nlmi) 0-based; qA (2 x nlmi);
        B: list of (nvar x msz) sparse or None; C_lin (nvar x nlin) sparse or None."""
        nlmi = len(AA)
        nvar = int(AA[0].shape[0]) if nlmi else int(C_lin.shape[0])
        keep = []
        cps, rvs, nzs = [], [], []
        for a in AA:
            cp, rv, nz = self._csc64(a)
            cps.append(cp); rvs.append(rv); nzs.append(nz)
        bcp = brv = bnz = None
        if B is not None and len(B) == nlmi and nlmi > 0:
            bcps, brvs, bnzs = [], [], []
            for b in B:
                cp, rv, nz = self._csc64(b)
                bcps.append(cp); brvs.append(rv); bnzs.append(nz)
            keep += [bcps, brvs, bnzs]
            bcp, brv, bnz = ptr_array(bcps), ptr_array(brvs), ptr_array(bnzs)
        ms = np.asarray(msizes, dtype=np.int64)
        sig = np.asfortranarray(np.asarray(sigmaA, dtype=np.int64) + 1)
        q = np.asfortranarray(np.asarray(qA, dtype=np.int64))
        nlin = 0
        lcp = lrv = lnz = None
        if C_lin is not None and C_lin.shape[1] > 0:
            nlin = int(C_lin.shape[1])
            lcp, lrv, lnz = self._csc64(C_lin)
        rc = self.lib.lrn_upload_model(
            self.h, nlmi, nvar, ptr(ms), ptr_array(cps), ptr_array(rvs), ptr_array(nzs), bcp, brv, bnz,
            ptr(sig), ptr(q), nlin, ptr(lcp), ptr(lrv), ptr(lnz))
        self._chk(rc, "lrn_upload_model")
        self.nvar, self.msizes, self.nlin = nvar, [int(x) for x in ms], nlin

    def synthetic_dense_model(self, msz, nvar, seed):
        self._chk(self.lib.lrn_synthetic_dense_model(self.h, int(msz), int(nvar), C.c_uint64(seed)),
                  "lrn_synthetic_dense_model")
        self.nvar, self.msizes, self.nlin = int(nvar), [int(msz)], 0

    def get_constraint(self, ilmi, k):
        m = self.msizes[ilmi]
        out = np.zeros((m, m), order="F")
        self._chk(self.lib.lrn_get_constraint(self.h, ilmi, int(k), ptr(out)), "lrn_get_constraint")
        return out

    # ------------------------------------------------------------------ scaling
    def set_scaling(self, ilmi, W, G=None):
        W = W if hasattr(W, "data_ptr") else f64(W)
        if G is not None and not hasattr(G, "data_ptr"):
            G = f64(G)
        self._chk(self.lib.lrn_set_scaling(self.h, ilmi, ptr(W), ptr(G)), "lrn_set_scaling")

    def set_lin(self, X_lin, S_lin_inv):
        self._chk(self.lib.lrn_set_lin(self.h, ptr(f64(X_lin)), ptr(f64(S_lin_inv))), "lrn_set_lin")

    def prepare_w(self, ilmi, X, S, want=True):
        """-> (info, dict(D,G,Gi,W,Si,DDsi)) ; outputs stay on the device when want=False."""
        m = self.msizes[ilmi]
        X = X if hasattr(X, "data_ptr") else f64(X)
        S = S if hasattr(S, "data_ptr") else f64(S)
        info = C.c_int(0)
        out = {}
        if want:
            for k_ in ("G", "Gi", "W", "Si"):
                out[k_] = np.zeros((m, m), order="F")
            out["D"] = np.zeros(m)
            out["DDsi"] = np.zeros(m)
        g = lambda k_: ptr(out[k_]) if want else None
        rc = self.lib.lrn_prepare_w(self.h, ilmi, ptr(X), ptr(S), g("D"), g("G"), g("Gi"), g("W"), g("Si"),
                                    g("DDsi"), C.byref(info))
        self._chk(rc, "lrn_prepare_w")
        return info.value, out

    # ------------------------------------------------------------------ Schur complement
    def schur_assemble(self, mode=0, want_H=False):
        H = np.zeros((self.nvar, self.nvar), order="F") if want_H else None
        self._chk(self.lib.lrn_schur_assemble(self.h, int(mode), ptr(H)), "lrn_schur_assemble")
        return H

    def schur_get(self):
        H = np.zeros((self.nvar, self.nvar), order="F")
        self._chk(self.lib.lrn_schur_get(self.h, ptr(H)), "lrn_schur_get")
        return H

    def schur_add_diag(self, eps):
        self._chk(self.lib.lrn_schur_add_diag(self.h, float(eps)), "lrn_schur_add_diag")

    def schur_factor(self):
        info = C.c_int(0)
        self._chk(self.lib.lrn_schur_factor(self.h, C.byref(info)), "lrn_schur_factor")
        return info.value

    def schur_solve(self, h):
        h = f64(h)
        x = np.zeros(self.nvar)
        self._chk(self.lib.lrn_schur_solve(self.h, ptr(h), ptr(x)), "lrn_schur_solve")
        return x

    def shard_doubles(self):
        return int(self.lib.lrn_schur_shard_doubles(self.h))

    def schur_is_partial_sum(self):
        """True when the last assembly left this rank's partial SUM of the whole Schur matrix (dense data through
        the Cholesky factor, world > 1): the exchange is then an all-reduce of `schur_export_full`."""
        return bool(self.lib.lrn_schur_is_partial_sum(self.h))

    def schur_plan(self, mode=0):
        """This rank's own view of the exchange the next assembly needs: 1 all-reduce of partial sums, 0 all-gather
        of Schur column blocks (lrn_schur_plan; the ranks agree on MIN and pin it with option "schur_plan")."""
        v = C.c_int(0)
        self._chk(self.lib.lrn_schur_plan(self.h, int(mode), C.byref(v)), "lrn_schur_plan")
        return int(v.value)

    def schur_export_full(self, buf):
        self._chk(self.lib.lrn_schur_export_full(self.h, ptr(buf)), "lrn_schur_export_full")

    def schur_import_full(self, buf):
        self._chk(self.lib.lrn_schur_import_full(self.h, ptr(buf)), "lrn_schur_import_full")

    def schur_export_shard(self, buf):
        self._chk(self.lib.lrn_schur_export_shard(self.h, ptr(buf)), "lrn_schur_export_shard")

    def schur_import_all(self, buf):
        self._chk(self.lib.lrn_schur_import_all(self.h, ptr(buf)), "lrn_schur_import_all")

    # ------------------------------------------------------------------ rhs / CG
    def make_rhs(self, Rp, RdS_list):
        Rp = f64(Rp)
        mats = [f64(M) for M in RdS_list]
        h = np.zeros(self.nvar)
        self._chk(self.lib.lrn_make_rhs(self.h, ptr(Rp), ptr_array(mats), ptr(h)), "lrn_make_rhs")
        return h

    def matvec(self, x):
        x = f64(x)
        y = np.zeros(self.nvar)
        self._chk(self.lib.lrn_matvec(self.h, ptr(x), ptr(y)), "lrn_matvec")
        return y

    def prec_setup(self, prec, erank, aamat):
        info = C.c_int(0)
        self._chk(self.lib.lrn_prec_setup(self.h, int(prec), int(erank), int(aamat), C.byref(info)),
                  "lrn_prec_setup")
        return info.value

    def prec_apply(self, x):
        x = f64(x)
        y = np.zeros(self.nvar)
        self._chk(self.lib.lrn_prec_apply(self.h, ptr(x), ptr(y)), "lrn_prec_apply")
        return y

    def pcg(self, h, tol, maxit=10000):
        h = f64(h)
        x = np.zeros(self.nvar)
        ec, it = C.c_int(0), C.c_int(0)
        self._chk(self.lib.lrn_pcg(self.h, ptr(h), float(tol), int(maxit), ptr(x), C.byref(ec), C.byref(it)),
                  "lrn_pcg")
        return x, ec.value, it.value

    # ------------------------------------------------------------------ probes / unit-test blocks
    def mfma_f64_peak(self):
        v = C.c_double(0.0)
        self._chk(self.lib.lrn_mfma_f64_peak(self.h, C.byref(v)), "lrn_mfma_f64_peak")
        return v.value

    def xcc_probe(self, nx, nz=1, hold_us=0):
        """XCC (XCD) id every workgroup of an (nx, 1, nz) grid ran on, as an (nz, nx) int32 array."""
        out = np.zeros((nz, nx), dtype=np.int32)
        self._chk(self.lib.lrn_xcc_probe(self.h, int(nx), int(nz), int(hold_us), out.ctypes.data_as(C.c_void_p)),
                  "lrn_xcc_probe")
        return out

    def hbm_copy_peak(self, nbytes=1 << 30):
        v = C.c_double(0.0)
        self._chk(self.lib.lrn_hbm_copy_peak(self.h, int(nbytes), C.byref(v)), "lrn_hbm_copy_peak")
        return v.value

    def dbg_gemm(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, Cin=None, flags=0, ksplit=1):
        A = f64(A); B = f64(B)
        M = A.shape[1] if transA else A.shape[0]
        K = A.shape[0] if transA else A.shape[1]
        N = B.shape[0] if transB else B.shape[1]
        Cm = np.zeros((M, N), order="F") if Cin is None else f64(Cin).copy(order="F")
        rc = self.lib.lrn_dbg_gemm(self.h, int(transA), int(transB), M, N, K, float(alpha), ptr(A), A.shape[0],
                                   ptr(B), B.shape[0], float(beta), ptr(Cm), Cm.shape[0], int(flags), int(ksplit))
        self._chk(rc, "lrn_dbg_gemm")
        return Cm

    def dbg_mfma_probe(self, A16x4, B4x16):
        A = np.require(A16x4, dtype=np.float64, requirements=["C"])
        B = np.require(B4x16, dtype=np.float64, requirements=["C"])
        D = np.zeros((16, 16), order="C")
        self._chk(self.lib.lrn_dbg_mfma_probe(self.h, ptr(A), ptr(B), ptr(D)), "lrn_dbg_mfma_probe")
        return D

    def dbg_potrf(self, A):
        A = f64(A).copy(order="F")
        info = C.c_int(0)
        self._chk(self.lib.lrn_dbg_potrf(self.h, A.shape[0], ptr(A), C.byref(info)), "lrn_dbg_potrf")
        return np.tril(A), info.value

    def dbg_potrs(self, A, b):
        A = f64(A); b = f64(b)
        x = np.zeros(A.shape[0])
        info = C.c_int(0)
        self._chk(self.lib.lrn_dbg_potrs(self.h, A.shape[0], ptr(A), ptr(b), ptr(x), C.byref(info)), "lrn_dbg_potrs")
        return x, info.value

    def dbg_trsm(self, A, B, trans=False):
        A = f64(A); B = f64(B).copy(order="F")
        info = C.c_int(0)
        self._chk(self.lib.lrn_dbg_trsm(self.h, A.shape[0], B.shape[1], int(trans), ptr(A), ptr(B), C.byref(info)),
                  "lrn_dbg_trsm")
        return B, info.value

    def dbg_svd_jacobi(self, A):
        A = f64(A)
        n = A.shape[0]
        U = np.zeros((n, n), order="F"); V = np.zeros((n, n), order="F"); s = np.zeros(n)
        sw = C.c_int(0)
        self._chk(self.lib.lrn_dbg_svd_jacobi(self.h, n, ptr(A), ptr(U), ptr(V), ptr(s), C.byref(sw)),
                  "lrn_dbg_svd_jacobi")
        return U, s, V, sw.value

    # ------------------------------------------------------------------ device-resident iterate
    def ip_set_c(self, ilmi, Cm):
        Cm = f64(Cm)
        self._chk(self.lib.lrn_ip_set_c(self.h, ilmi, ptr(Cm)), "lrn_ip_set_c")

    def ip_set_iterate(self, ilmi, X, S):
        X = f64(X); S = f64(S)
        self._chk(self.lib.lrn_ip_set_iterate(self.h, ilmi, ptr(X), ptr(S)), "lrn_ip_set_iterate")

    def ip_get_iterate(self, ilmi):
        m = self.msizes[ilmi]
        X = np.zeros((m, m), order="F"); S = np.zeros((m, m), order="F")
        self._chk(self.lib.lrn_ip_get_iterate(self.h, ilmi, ptr(X), ptr(S)), "lrn_ip_get_iterate")
        return X, S

    def dbg_get_block(self, ilmi, name):
        """One array of the resident state of a block (see lrn_dbg_get_block); returns (array, eigen_free_flag)."""
        m = self.msizes[ilmi]
        out = np.zeros(m) if name in ("D", "DDsi") else np.zeros((m, m), order="F")
        flag = C.c_int(0)
        self._chk(self.lib.lrn_dbg_get_block(self.h, ilmi, name.encode(), ptr(out), C.byref(flag)), "lrn_dbg_get_block")
        return out, flag.value

    def ip_add_diag(self, ilmi, which, eps):
        self._chk(self.lib.lrn_ip_add_diag(self.h, ilmi, int(which), float(eps)), "lrn_ip_add_diag")

    def ip_prepare_w(self, ilmi):
        info = C.c_int(0)
        self._chk(self.lib.lrn_ip_prepare_w(self.h, ilmi, C.byref(info)), "lrn_ip_prepare_w")
        return info.value

    def _vec_out(self, fn, name, *args):
        out = np.zeros(self.nvar)
        self._chk(fn(self.h, *args, ptr(out)), name)
        return out

    def ip_aa_x(self):
        return self._vec_out(self.lib.lrn_ip_aa_x, "lrn_ip_aa_x")

    def ip_residual_d(self, y):
        y = f64(y)
        self._chk(self.lib.lrn_ip_residual_d(self.h, ptr(y)), "lrn_ip_residual_d")

    def ip_rhs_pred(self):
        return self._vec_out(self.lib.lrn_ip_rhs_pred, "lrn_ip_rhs_pred")

    def ip_rhs_pred2(self):
        """(AA*vec(X), makeRHS term) with dense constraint data read once."""
        aax = np.zeros(self.nvar); out = np.zeros(self.nvar)
        self._chk(self.lib.lrn_ip_rhs_pred2(self.h, ptr(aax), ptr(out)), "lrn_ip_rhs_pred2")
        return aax, out

    def ip_rhs_corr(self, sigma_mu):
        return self._vec_out(self.lib.lrn_ip_rhs_corr, "lrn_ip_rhs_corr", C.c_double(sigma_mu))

    def ip_find_step(self, predict, sigma_mu, tau, dely):
        dely = f64(dely)
        nl = max(1, len(self.msizes))
        a = np.zeros(nl); b = np.zeros(nl)
        self._chk(self.lib.lrn_ip_find_step(self.h, int(predict), float(sigma_mu), float(tau), ptr(dely), ptr(a), ptr(b)),
                  "lrn_ip_find_step")
        return a[:len(self.msizes)], b[:len(self.msizes)]

    def ip_update(self, predict, alpha, beta):
        nl = max(1, len(self.msizes))
        a = np.zeros(nl); b = np.zeros(nl); tr = np.zeros(nl)
        a[:len(np.atleast_1d(alpha))] = alpha; b[:len(np.atleast_1d(beta))] = beta
        self._chk(self.lib.lrn_ip_update(self.h, int(predict), ptr(a), ptr(b), ptr(tr)), "lrn_ip_update")
        return tr[:len(self.msizes)]

    def ip_stats(self):
        nl = len(self.msizes)
        out = np.zeros(5 * max(1, nl))
        self._chk(self.lib.lrn_ip_stats(self.h, ptr(out)), "lrn_ip_stats")
        return out[:5 * nl].reshape(nl, 5)

    def dbg_lanczos(self, M, k, vectors=True):
        M = f64(M)
        n = M.shape[0]
        lam = np.zeros(max(k, 1)); U = np.zeros((n, max(k, 1)), order="F")
        lmin = C.c_double(0.0); tr = C.c_double(0.0); st = C.c_int(0)
        self._chk(self.lib.lrn_dbg_lanczos(self.h, n, int(k), ptr(M), ptr(lam), ptr(U) if vectors else None,
                                           C.byref(lmin), C.byref(tr), C.byref(st)), "lrn_dbg_lanczos")
        return lam[:k], U[:, :k], lmin.value, tr.value, st.value

    def dbg_eigmin(self, M, certified=False):
        """certified=False: the plain Lanczos Ritz value and its step count; True: the Cholesky-certified
        value the step-length rule and the DIMACS errors use (steps = 0)."""
        M = f64(M)
        lam = C.c_double(0.0); st = C.c_int(0)
        self._chk(self.lib.lrn_dbg_eigmin(self.h, M.shape[0], ptr(M), C.byref(lam), None if certified else C.byref(st)),
                  "lrn_dbg_eigmin")
        return lam.value, st.value
