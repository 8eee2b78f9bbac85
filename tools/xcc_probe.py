"""Where do workgroups run?  For grids shaped like the assembly GEMMs, compare the XCC id the hardware reports
(HW_REG_XCC_ID) with the `flat workgroup id mod 8` rule the tile swizzle of gemm_f64.hip assumes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd

dev = loraine_jl_amd.Device(0)
for nx, nz, hold in ((528, 64, 50), (136, 256, 50), (528, 16, 0), (4096, 1, 20), (530, 8, 20)):
    x = dev.xcc_probe(nx, nz, hold)
    flat = (np.arange(nx)[None, :] + nx * np.arange(nz)[:, None])
    agree = float(np.mean(x == (flat % 8)))
    # best constant rotation
    rot = max(range(8), key=lambda r: np.mean(x == ((flat + r) % 8)))
    agree_rot = float(np.mean(x == ((flat + rot) % 8)))
    print(f"grid ({nx},1,{nz}) hold {hold} us: xcc values {sorted(set(x.ravel().tolist()))}, "
          f"== flat%8: {agree:.4f}, best rotation {rot}: {agree_rot:.4f}; first row head {x[0, :24].tolist()}", flush=True)
    if nz > 1:
        print("   second row head", x[1, :24].tolist(), " per-xcc counts", np.bincount(x.ravel(), minlength=8).tolist(), flush=True)
