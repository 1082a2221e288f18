"""Test helpers: device buffers over the C-ABI and Montgomery conversions."""
import ctypes as C

import numpy as np

P = 2013265921
R = (1 << 32) % P
RINV = pow(R, -1, P)


def to_monty(a):
    return ((np.asarray(a, dtype=np.uint64) * np.uint64(R)) % np.uint64(P)).astype(np.uint32)


def from_monty(a):
    return ((np.asarray(a, dtype=np.uint64) * np.uint64(RINV)) % np.uint64(P)).astype(np.uint32)


def bitrev_perm(logn):
    n = 1 << logn
    idx = np.arange(n, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint32)
    for b in range(logn):
        out |= ((idx >> b) & 1) << (logn - 1 - b)
    return out


class DevBuf:
    def __init__(self, gpu, nbytes):
        self.gpu, self.nbytes = gpu, nbytes
        self.ptr = C.c_void_p()
        gpu.check(gpu.lib.zksp_dev_malloc(gpu.h, max(nbytes, 4), C.byref(self.ptr)))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.gpu.check(self.gpu.lib.zksp_dev_upload(self.gpu.h, self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.zeros(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.gpu.check(self.gpu.lib.zksp_dev_download(self.gpu.h, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.gpu.lib.zksp_dev_free(self.gpu.h, self.ptr)
            self.ptr = None


class Gpu:
    """One HIP client + raw access to the kernel-level C-ABI entry points."""

    def __init__(self, zk, **kw):
        self.zk = zk
        self.client = zk.ProverClient(device=0, **kw)
        self.lib = self.client._lib
        self.h = self.client._h

    def check(self, rc):
        if rc:
            raise RuntimeError(f"zksp rc={rc}: {self.client.last_error()}")

    def buf(self, arr=None, nbytes=None):
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            return DevBuf(self, arr.nbytes).upload(arr)
        return DevBuf(self, nbytes)

    # --- kernel wrappers taking / returning canonical numpy arrays ---
    def permute(self, states):
        st = np.ascontiguousarray(states, dtype=np.uint32).reshape(-1, 16)
        b = self.buf(to_monty(st))
        self.check(self.lib.zksp_hip_poseidon2_permute(self.h, b.ptr, st.shape[0]))
        out = from_monty(b.download(np.uint32, st.shape))
        b.free()
        return out

    def merkle_commit(self, mat):
        mat = np.ascontiguousarray(mat, dtype=np.uint32)
        w, n = mat.shape
        logn = n.bit_length() - 1
        m = self.buf(to_monty(mat))
        t = self.buf(nbytes=(2 * n - 1) * 32)
        self.check(self.lib.zksp_hip_merkle_commit(self.h, m.ptr, w, logn, t.ptr))
        out = from_monty(t.download(np.uint32, (2 * n - 1, 8)))
        m.free(); t.free()
        return out

    def lde(self, cols, in_shift=1):
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        ncols, h = cols.shape
        logh = h.bit_length() - 1
        i = self.buf(to_monty(cols))
        cf = self.buf(nbytes=cols.nbytes)
        o = self.buf(nbytes=2 * cols.nbytes)
        self.check(self.lib.zksp_hip_lde(self.h, i.ptr, logh, ncols, in_shift, cf.ptr, o.ptr))
        coefs_br = from_monty(cf.download(np.uint32, (ncols, h)))
        lde = from_monty(o.download(np.uint32, (ncols, 2, h)))
        i.free(); cf.free(); o.free()
        br = bitrev_perm(logh)
        coefs = np.zeros_like(coefs_br)
        coefs[:, br] = coefs_br  # coefs_br[pos] holds coefficient bitrev(pos)
        return lde, coefs

    def keccak_trace(self, states, logh):
        st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 25)
        s = self.buf(st)
        t = self.buf(nbytes=2633 * (1 << logh) * 4)
        self.check(self.lib.zksp_hip_keccak_trace(self.h, s.ptr, st.shape[0], logh, t.ptr))
        out = from_monty(t.download(np.uint32, (2633, 1 << logh)))
        s.free(); t.free()
        return out

    def keccak_quotient(self, lde, lde_p, alpha, gamma, beta, cum_sum):
        lde = np.ascontiguousarray(lde, dtype=np.uint32)
        lde_p = np.ascontiguousarray(lde_p, dtype=np.uint32)
        w, two, h = lde.shape
        logh = h.bit_length() - 1
        l = self.buf(to_monty(lde))
        lp = self.buf(to_monty(lde_p))
        q = self.buf(nbytes=8 * h * 4)
        ch = np.ascontiguousarray(np.concatenate([alpha, gamma, beta, cum_sum]), dtype=np.uint32)
        self.check(self.lib.zksp_hip_keccak_quotient(self.h, l.ptr, lp.ptr, logh, ch.ctypes.data_as(C.c_void_p), q.ptr))
        out = from_monty(q.download(np.uint32, (8, h)))
        l.free(); lp.free(); q.free()
        return out

    def bus_perm_trace(self, trace, gamma, beta):
        trace = np.ascontiguousarray(trace, dtype=np.uint32)
        w, h = trace.shape
        logh = h.bit_length() - 1
        t = self.buf(to_monty(trace))
        phi = self.buf(nbytes=4 * h * 4)
        cum = self.buf(nbytes=16)
        gb = np.ascontiguousarray(np.concatenate([gamma, beta]), dtype=np.uint32)
        self.check(self.lib.zksp_hip_bus_perm_trace(self.h, t.ptr, logh, gb.ctypes.data_as(C.c_void_p), phi.ptr, cum.ptr))
        out = from_monty(phi.download(np.uint32, (4, h))), from_monty(cum.download(np.uint32, (4,)))
        t.free(); phi.free(); cum.free()
        return out

    def fri_fold(self, layer, shift_k, beta):
        layer = np.ascontiguousarray(layer, dtype=np.uint32)
        _, hk, _ = layer.shape
        loghk = hk.bit_length() - 1
        i = self.buf(to_monty(layer))
        o = self.buf(nbytes=layer.nbytes // 2)
        b = np.ascontiguousarray(beta, dtype=np.uint32)
        self.check(self.lib.zksp_hip_fri_fold(self.h, i.ptr, loghk, shift_k, b.ctypes.data_as(C.c_void_p), o.ptr))
        out = from_monty(o.download(np.uint32, (2, hk // 2, 4)))
        i.free(); o.free()
        return out

    def microbench(self, which):
        g = C.c_double()
        self.check(self.lib.zksp_hip_microbench(self.h, which, C.byref(g)))
        return g.value
