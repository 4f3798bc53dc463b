"""Device memory with NOTHING mapped on either side of it (test infrastructure).

HIP's virtual-memory calls: reserve an address range of three pieces, back and map only the middle one.  A buffer handed out by
`tail(n)` ends exactly where the mapping ends, one handed out by `head(n)` starts exactly where it starts: a kernel (or a copy
engine) that touches a single byte past the caller's buffer - a vector load that takes a whole pixel where only its first bytes
exist, a row fetched once too often - takes a memory access fault, which ends the process.  tests/test_gpu_guard_pages.py runs
every scenario in a child process and reads its exit status; parity tests on ordinary allocations cannot see such reads (the
allocator's blocks are padded, and it took the 2 263rd random frame of a soak to place one against an unmapped page).
"""
import ctypes as C


class _Location(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class _AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class _AllocationProp(C.Structure):          # hipMemAllocationProp (hip_runtime_api.h)
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", _Location), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", _AllocFlags)]


class _AccessDesc(C.Structure):
    _fields_ = [("location", _Location), ("flags", C.c_int)]


_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        _hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(_AllocationProp), C.c_ulonglong]
        _hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        _hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_AccessDesc), C.c_size_t]
        _hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(C.c_size_t), C.POINTER(_AllocationProp), C.c_int]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
        _hip.hipMemRelease.argtypes = [C.c_void_p]
        _hip.hipMemAddressFree.argtypes = [C.c_void_p, C.c_size_t]
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hipError {rc}")


class GuardedRegion:
    """`nbytes` (rounded up to the mapping granularity) of device memory between two unmapped address ranges."""

    def __init__(self, nbytes, device=0):
        h = hip()
        prop = _AllocationProp()
        prop.type = 1                        # hipMemAllocationTypePinned
        prop.location = _Location(1, device)  # hipMemLocationTypeDevice
        gran = C.c_size_t()
        _ok(h.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 0), "hipMemGetAllocationGranularity")
        self.granularity = gran.value
        self.size = (max(nbytes, 1) + self.granularity - 1) // self.granularity * self.granularity
        base = C.c_void_p()
        self.reserved = self.size + 2 * self.granularity
        _ok(h.hipMemAddressReserve(C.byref(base), self.reserved, 0, None, 0), "hipMemAddressReserve")
        self.base = base.value
        self.start = self.base + self.granularity
        self.handle = C.c_void_p()
        _ok(h.hipMemCreate(C.byref(self.handle), self.size, C.byref(prop), 0), "hipMemCreate")
        _ok(h.hipMemMap(self.start, self.size, 0, self.handle, 0), "hipMemMap")
        acc = _AccessDesc(_Location(1, device), 3)      # hipMemAccessFlagsProtReadWrite
        _ok(h.hipMemSetAccess(self.start, self.size, C.byref(acc), 1), "hipMemSetAccess")
        _ok(h.hipMemset(self.start, 0xA5, self.size), "hipMemset")
        self.end = self.start + self.size

    def tail(self, nbytes):
        """Address of a buffer of `nbytes` whose last byte is the last mapped byte."""
        assert 0 < nbytes <= self.size
        return self.end - nbytes

    def head(self, nbytes):
        """Address of a buffer of `nbytes` whose first byte is the first mapped byte."""
        assert 0 < nbytes <= self.size
        return self.start

    def upload(self, dst, array):
        import numpy as np
        a = np.ascontiguousarray(array)
        assert self.start <= dst and dst + a.nbytes <= self.end
        _ok(hip().hipMemcpy(dst, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")

    def download(self, src, nbytes):
        import numpy as np
        assert self.start <= src and src + nbytes <= self.end
        out = np.empty(nbytes, dtype=np.uint8)
        _ok(hip().hipMemcpy(out.ctypes.data, src, nbytes, 2), "hipMemcpy D2H")
        return out

    def close(self):
        if self.handle:
            h = hip()
            h.hipDeviceSynchronize()
            h.hipMemUnmap(self.start, self.size)
            h.hipMemRelease(self.handle)
            h.hipMemAddressFree(self.base, self.reserved)
            self.handle = None
