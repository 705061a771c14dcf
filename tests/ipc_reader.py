"""Child process of tests/test_gpu_parity.py::test_hip_meshes_exported_to_another_process: a 'renderer' that knows nothing of the
library - it opens the two inter-process handles (hipIpcOpenMemHandle) and prints a digest of what it finds there.
Usage: python tests/ipc_reader.py <verts handle hex> <indices handle hex> <n_verts> <n_indices>"""
import ctypes as C
import hashlib
import sys

try:
    import torch  # noqa: F401  (the exporting test process runs on the HIP runtime PyTorch ships; a handle is only good for the runtime that made it)
except ImportError:
    pass


def main():
    vh, ih, nv, ni = bytes.fromhex(sys.argv[1]), bytes.fromhex(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    hip = C.CDLL("libamdhip64.so")

    class Handle(C.Structure):
        _fields_ = [("reserved", C.c_char * 64)]

    hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipSetDevice(0) == 0
    out = []
    for raw, nbytes in ((vh, nv * 48), (ih, ni * 4)):
        h = Handle()
        C.memmove(C.byref(h), raw, 64)
        p = C.c_void_p()
        rc = hip.hipIpcOpenMemHandle(C.byref(p), h, 1)  # hipIpcMemLazyEnablePeerAccess
        assert rc == 0, "hipIpcOpenMemHandle: %d" % rc
        buf = (C.c_uint8 * nbytes)()
        assert hip.hipMemcpy(buf, p, nbytes, 2) == 0
        out.append(hashlib.sha256(bytes(buf)).hexdigest())
        assert hip.hipIpcCloseMemHandle(p) == 0
    print("ipc digest %s %s" % (out[0], out[1]))


if __name__ == "__main__":
    main()
