"""Eigen-free NT scaling against the SVD route on the device, on synthetic iterates (X, S) off the central path:
W and Si of both routes, WSW = X, the Newton-Schulz step count and the time of lrn_ip_prepare_w.
Usage: python tools/nt_probe.py [msz ...]   (LRN_OPTS=key=value,... sets library options)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import loraine_jl_amd
from loraine_jl_amd import Device

def spd(n, cond, rng):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.exp(rng.uniform(0, np.log(cond), n))
    return (Q * ev) @ Q.T

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [200, 800, 2000]
    dev = Device(0)
    dev.set_option("profile", 1)
    for kv in filter(None, os.environ.get("LRN_OPTS", "").split(",")):
        k, v = kv.split("="); dev.set_option(k, float(v))
    rng = np.random.default_rng(7)
    for n in sizes:
        nvar = 4
        AA = sp.csc_matrix((np.ones(nvar), (np.arange(nvar), np.arange(nvar) * (n + 1))), shape=(nvar, n * n))
        dev.upload_model([AA], np.arange(nvar, dtype=np.int64).reshape(-1, 1), np.zeros((2, 1), dtype=np.int64), [n])
        X = spd(n, float(os.environ.get("COND_X", "1e6")), rng)
        # S near mu X^-1 times a perturbation of moderate condition: cond(K) ~ 1e2
        E = spd(n, float(os.environ.get("COND_K", "1e2")), rng)
        ev, Q = np.linalg.eigh(X)
        Lx = Q / np.sqrt(ev)                      # X^-1 = Lx Lx'
        S = Lx @ E @ Lx.T
        S = (S + S.T) / 2
        dev.ip_set_c(0, np.eye(n))
        out = {}
        for mode in (0, 1):
            dev.set_option("nt_mode", mode)
            dev.ip_set_iterate(0, X, S)
            dev.ip_prepare_w(0)                       # warm-up (allocations)
            dev.ip_set_iterate(0, X, S)
            dev.reset_timing()
            t = time.perf_counter(); info = dev.ip_prepare_w(0); wall = (time.perf_counter() - t) * 1e3
            W, flag = dev.dbg_get_block(0, "W")
            Si, _ = dev.dbg_get_block(0, "Si")
            out[mode] = (W, Si)
            r = np.linalg.norm(W @ S @ W - X) / np.linalg.norm(X)
            print("msz %5d mode %d info %d flag %d: prepare_w %.2f ms (wall %.2f)  [chol %.2f gemm %.2f ns %.2f svd %.2f] ns_steps %d  |WSW-X|/|X| %.2e"
                  % (n, mode, info, flag, dev.timing("prepare_w"), wall, dev.timing("prepw_chol"), dev.timing("prepw_gemm"),
                     dev.timing("prepw_ns"), dev.timing("prepw_svd"), dev.count("ns_steps"), r), flush=True)
        # reference W in NumPy: W = L_X K^-1/2 L_X' through eigh of K
        LXn = np.linalg.cholesky(X); Kn = LXn.T @ S @ LXn; ev, V = np.linalg.eigh((Kn + Kn.T) / 2)
        Wn = LXn @ ((V / np.sqrt(ev)) @ V.T) @ LXn.T
        print("   cond(K) %.1e   W(svd) vs numpy %.2e   W(ns) vs numpy %.2e" % (ev[-1] / ev[0],
              np.linalg.norm(out[0][0] - Wn) / np.linalg.norm(Wn), np.linalg.norm(out[1][0] - Wn) / np.linalg.norm(Wn)))
        dW = np.linalg.norm(out[0][0] - out[1][0]) / np.linalg.norm(out[0][0])
        dS = np.linalg.norm(out[0][1] - out[1][1]) / np.linalg.norm(out[0][1])
        print("   W(ns) vs W(svd) %.2e   Si %.2e" % (dW, dS), flush=True)

main()
